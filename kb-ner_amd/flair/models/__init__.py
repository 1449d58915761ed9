from .sequence_tagger_model import FastSequenceTagger, SequenceTagger  # noqa: F401
from .language_model import LanguageModel  # noqa: F401
