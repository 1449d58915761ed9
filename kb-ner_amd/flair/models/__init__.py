from .sequence_tagger_model import FastSequenceTagger, SequenceTagger  # noqa: F401
