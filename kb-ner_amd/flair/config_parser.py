"""YAML -> corpus / embeddings / model.  Behavioural reference (restated): flair/config_parser.py -- ConfigParser.__init__
(:28-115: `targets`, `<target>.Corpus` split on ':', ColumnCorpus-<NAME> entries built from `<target>[<entry>]` kwargs
:331-336, tag dictionary loaded from `<target>.tag_dictionary` if the file exists else made from the corpus and saved
:109-115), create_embeddings (:145-184: `ClassName-idx` keys -> getattr(embeddings, ClassName)(**kwargs)), create_model
(:185-234: getattr(models, classname)(**kwargs, embeddings, tag_type, tag_dictionary, target_languages, config)),
create_student (:236), get_target_path (:602).  Only ColumnCorpus entries are supported (named corpora need downloads)."""
import copy
import logging
from pathlib import Path

import flair.datasets as datasets
import flair.embeddings as Embeddings
import flair.models as models
from flair.data import Dictionary
from flair.list_data import ListCorpus
from flair.utils.params import Params  # noqa: F401

log = logging.getLogger("flair")


class ConfigParser:
    def __init__(self, config, all=False, zero_shot=False, other_shot=False, predict=False, save_embedding=False):
        if all or zero_shot or other_shot or predict:
            raise NotImplementedError("only the plain training corpus selection is on the hot path")
        self.config = config
        self.mini_batch_size = self.config["train"]["mini_batch_size"]
        self.target = self.get_target
        self.tag_type = self.target
        if save_embedding:
            self.corpus, self.tokens, self.tag_dictionary, self.num_corpus = None, None, {}, None
            return
        self.corpus = self.get_corpus
        self.tokens = None
        self.corpus_list = self.config[self.target]["Corpus"].split(":")
        td = self.config[self.target].get("tag_dictionary")
        if td and Path(td).exists():
            self.tag_dictionary = Dictionary.load_from_file(td)
        else:
            # every rank derives the same dictionary from the same corpus; only rank 0 writes it, atomically (a reader never
            # sees a half-written pickle)
            self.tag_dictionary = self.corpus.make_tag_dictionary(tag_type=self.target)
            import os
            if td and int(os.environ.get("RANK", "0")) == 0:
                Path(td).parent.mkdir(parents=True, exist_ok=True)
                tmp = "%s.tmp.%d" % (td, os.getpid())
                self.tag_dictionary.save(tmp)
                os.replace(tmp, td)
        log.info(self.tag_dictionary.item2idx)
        self.num_corpus = len(self.corpus.targets)
        log.info(self.corpus)

    @property
    def get_target(self):
        targets = self.config.get("targets").split(":")
        if len(targets) > 1:
            log.info("Warning! Not support multitask now!")
        return targets[0]

    @property
    def get_corpus(self):
        parts = {"train": [], "dev": [], "test": []}
        names = self.config[self.target]["Corpus"].split(":")
        for entry in names:
            cls_name = entry.split("-", 1)[0]
            if cls_name != "ColumnCorpus":
                raise NotImplementedError("corpus %r: only ColumnCorpus-<NAME> entries are supported (named flair corpora "
                                          "need downloads / private paths)" % entry)
            ds = datasets.ColumnCorpus(**self.config[self.target][entry])
            parts["train"].append(ds.train)
            parts["dev"].append(ds.dev)
            parts["test"].append(ds.test)
        return ListCorpus(**parts, targets=names)

    def create_embeddings(self, embeddings: dict):
        built = []
        for key, kw in embeddings.items():
            cls = getattr(Embeddings, key.split("-")[0], None)
            if cls is None:
                raise NotImplementedError("embedding class %s is outside the hot path" % key.split("-")[0])
            built.append(cls(**kw) if isinstance(kw, dict) else cls())
        return Embeddings.StackedEmbeddings(embeddings=built), None, None, None, None

    def create_model(self, config=None, pretrained=False, is_student=False, crf=True):
        config = self.config if config is None else config
        embeddings, word_map, char_map, lemma_map, postag_map = self.create_embeddings(config["embeddings"])
        classname = list(config["model"].keys())[0]
        kwargs = copy.deepcopy(config["model"][classname])
        if not crf:
            kwargs["use_crf"] = False
        kwargs.update(embeddings=embeddings, tag_type=self.target, tag_dictionary=self.tag_dictionary)
        if not pretrained:
            kwargs["target_languages"] = self.num_corpus
        tagger = getattr(models, classname)(**kwargs, config=config)
        tagger.word_map, tagger.char_map, tagger.lemma_map, tagger.postag_map = word_map, char_map, lemma_map, postag_map
        if pretrained:
            base = Path(config["target_dir"]) / config["model_name"]
            for name in ("best-model.pt", "final-model.pt"):
                if (base / name).exists():
                    log.info("Loading pretraining %s", name)
                    tagger = tagger.load(base / name)
                    break
            else:
                raise FileNotFoundError(str(base) + " not exist!")
        tagger.use_bert = any("bert" in k.lower() for k in config["embeddings"])
        return tagger

    def create_student(self, nocrf=False):
        return self.create_model(self.config, pretrained=self.load_pretrained(self.config), is_student=True, crf=not nocrf)

    def create_teachers(self, is_professor=False):
        """config_parser.py:242-253: one teacher per training corpus, built from `<target>[<corpus>].train_config` (its own
        YAML: embeddings + model + target_dir / model_name) and loaded from that run's best-model.pt; it teaches that corpus"""
        if is_professor:
            raise NotImplementedError("professors (train_with_professor) are outside the hot path")
        teachers = []
        for corpus in self.corpus_list:
            config = Params.from_file(self.config[self.target][corpus]["train_config"])
            teacher = self.create_model(config, pretrained=True)
            teacher.targets = {corpus}
            self._as_frozen_teacher(teacher)
            teachers.append(teacher)
        return teachers

    @staticmethod
    def _as_frozen_teacher(teacher):
        """a teacher only ever labels the training set (forward passes): its gradient buffer -- as large as its parameters -- is
        given back at once (the reference moves its teachers to the CPU after labelling, finetune_trainer.py:633-636)"""
        drop = getattr(teacher, "drop_gradients", None)
        if drop is not None:
            drop()

    def create_teachers_list(self, is_professor=False):
        """config_parser.py:255-274 (`is_teacher_list: true`): `<target>.teachers` maps a teacher's YAML to the ':'-joined corpora
        it teaches; teachers of corpora this run does not train on are skipped"""
        if is_professor:
            raise NotImplementedError("professors (train_with_professor) are outside the hot path")
        teachers = []
        configs = self.config[self.target]["teachers"]
        for filename in configs:
            corpus_target = set(configs[filename].split(":"))
            if not (set(self.corpus.targets) & corpus_target):
                continue
            teacher = self.create_model(Params.from_file(filename), pretrained=True)
            teacher.targets = corpus_target
            self._as_frozen_teacher(teacher)
            teachers.append(teacher)
        return teachers

    def load_pretrained(self, config):
        return bool(self.config.get("load_pretrained", False))

    @property
    def get_target_path(self):
        return Path(self.config["target_dir"]) / self.config["model_name"]
