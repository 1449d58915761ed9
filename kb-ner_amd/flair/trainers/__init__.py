from .finetune_trainer import ModelFinetuner  # noqa: F401
from .reinforcement_trainer import ReinforcementTrainer  # noqa: F401
