"""ReinforcementTrainer surface for BASELINE config 5 (ACE-style stacked embeddings, INFERENCE only).

Behavioural reference (restated): flair/trainers/reinforcement_trainer.py -- constructor keywords (:61-84) and what it does to
the model (:97-101: `use_rl`, `embedding_selector`), train() keywords (:273-322).  train.py selects this class by the YAML's
`trainer: ReinforcementTrainer` (train.py:104-131) and, in --parse / --test mode, only needs the constructed trainer's
`.corpus`, `.assign_ext_context_doc`, `.final_test`; it then loads `training_state.pt` and sets `student.selection =
best_action` itself (train.py:213-253).  ACE's controller TRAINING (the RL search over embedding subsets) is out of scope
(BASELINE config 5 is inference-only): train() raises.

Note: the shipped ACE YAML passes `assign_doc_for_ext_context` and string optimiser names to this constructor, which the
reference's own signature does not accept (its **kwargs is commented out, :83) -- the mirror accepts them."""
import logging

from .finetune_trainer import ModelFinetuner, _warn_unknown

log = logging.getLogger("flair")


class ReinforcementTrainer(ModelFinetuner):
    def __init__(self, model, teachers, corpus, optimizer=None, controller_optimizer=None, controller_learning_rate: float = 0.1,
                 epoch: int = 0, distill_mode=False, optimizer_state: dict = None, scheduler_state: dict = None,
                 use_tensorboard: bool = False, language_resample=False, config=None, is_test: bool = False,
                 direct_upsample_rate: int = -1, down_sample_amount: int = -1, sentence_level_batch: bool = False,
                 dev_sample: bool = False, assign_doc_id: bool = False, train_with_doc: bool = False,
                 pretrained_file_dict: dict = None, sentence_level_pretrained_data: bool = False,
                 assign_doc_for_ext_context: bool = False, **kwargs):
        _warn_unknown("ReinforcementTrainer", kwargs)
        super().__init__(model, teachers, corpus, optimizer=optimizer, epoch=epoch, optimizer_state=optimizer_state,
                         scheduler_state=scheduler_state, use_tensorboard=use_tensorboard, distill_mode=distill_mode, config=config,
                         is_test=is_test, language_resample=language_resample, direct_upsample_rate=direct_upsample_rate,
                         down_sample_amount=down_sample_amount, sentence_level_batch=sentence_level_batch,
                         assign_doc_id=assign_doc_id, train_with_doc=train_with_doc, pretrained_file_dict=pretrained_file_dict,
                         sentence_level_pretrained_data=sentence_level_pretrained_data,
                         assign_doc_for_ext_context=assign_doc_for_ext_context)
        self.controller_learning_rate = controller_learning_rate
        self.controller_optimizer = controller_optimizer
        # reinforcement_trainer.py:97-101: the tagger multiplies every embedding's features by `selection[idx]` in forward()
        self.model.use_rl = True
        self.model.embedding_selector = True
        if getattr(self.model, "selection", None) is None:
            n = len(self.model.embeddings.embeddings) if hasattr(self.model.embeddings, "embeddings") else 1
            self.model.selection = [1] * n   # all embeddings on until train.py loads best_action

    def train(self, base_path, learning_rate: float = 5e-5, mini_batch_size: int = 32, eval_mini_batch_size: int = None,
              max_epochs: int = 100, max_episodes: int = 10, anneal_factor: float = 0.5, patience: int = 10,
              min_learning_rate: float = 5e-9, train_with_dev: bool = False, macro_avg: bool = True, monitor_train: bool = False,
              monitor_test: bool = False, embeddings_storage_mode: str = "cpu", checkpoint: bool = False,
              save_final_model: bool = True, anneal_with_restarts: bool = False, shuffle: bool = True,
              true_reshuffle: bool = False, param_selection_mode: bool = False, num_workers: int = 4, sampler=None,
              use_amp: bool = False, amp_opt_level: str = "O1", max_epochs_without_improvement=30, warmup_steps: int = 0,
              use_warmup: bool = True, gradient_accumulation_steps: int = 1, lr_rate: int = 1, decay: float = 0.75,
              decay_steps: int = 5000, sort_data: bool = True, fine_tune_mode: bool = False, debug: bool = False,
              min_freq: int = -1, min_lemma_freq: int = -1, min_pos_freq: int = -1, rootschedule: bool = False,
              freezing: bool = False, log_reward: bool = False, sqrt_reward: bool = False, controller_momentum: float = 0.0,
              discount: float = 0.5, curriculum_file=None, random_search=False, continue_training=False, old_reward=False,
              one_by_one: bool = False, select_model_by_macro: bool = False, **kwargs):
        raise NotImplementedError("ACE controller training (the reinforcement search over embedding subsets) is outside this "
                                  "build's scope: BASELINE config 5 is inference-only -- use --parse / --test with a trained "
                                  "model directory (best-model.pt + training_state.pt)")
