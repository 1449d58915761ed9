from .finetune_trainer import ModelFinetuner  # noqa: F401
