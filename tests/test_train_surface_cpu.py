"""CPU: the mirror satisfies the API surface the reference's entry script exercises (north_star: "train.py drops in unchanged").

tests/golden/train_surface.json is produced by oracle/gen_train_surface.py from the `ast` of /root/reference/train.py and the
keyword sets of /root/reference/config/*.yaml: every `from flair... import`, every attribute train.py touches on the student /
trainer / corpus / config objects, every keyword it passes.  This test walks that fixture against kb-ner_amd/flair statically
(signatures + class sources; constructing a tagger needs the GPU -- tests/test_gpu_flair_e2e.py::test_train_surface_live does
the same walk on live objects)."""
import ast
import importlib
import inspect
import json
import os
import textwrap

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SURFACE = json.load(open(os.path.join(HERE, "golden", "train_surface.json")))

# Names train.py mentions that belong to paths SURVEY.md §8 marks out of scope; each with the reason.  Anything NOT listed here
# must exist.
OUT_OF_SCOPE = {
    ("datasets_module", "UniversalDependenciesCorpus"): "dependency parsing branch (student.tag_type == 'dependency'), SURVEY §2.1",
    ("embedding", "ee"): "ELMo needs external weight files (SURVEY §8f-1: 'can stay stubbed'); guarded by `'elmo' in embedding.name`",
    ("embedding", "is_hit_elmo"): "same ELMo-only branch (train.py:228)",
}


def _class_attrs(cls):
    """names assigned as `self.x = ...` anywhere in the class hierarchy + class attributes / methods / properties"""
    names = set()
    for c in cls.__mro__:
        if c is object:
            continue
        names.update(vars(c).keys())
        try:
            tree = ast.parse(textwrap.dedent(inspect.getsource(c)))
        except (OSError, TypeError):
            continue
        for node in ast.walk(tree):
            if isinstance(node, ast.Attribute) and isinstance(node.ctx, ast.Store) and isinstance(node.value, ast.Name) \
                    and node.value.id == "self":
                names.add(node.attr)
    return names


def _explicit_params(fn):
    sig = inspect.signature(fn)
    return {n for n, p in sig.parameters.items() if p.kind in (p.POSITIONAL_OR_KEYWORD, p.KEYWORD_ONLY)}


def _roles():
    import flair
    import flair.datasets
    import flair.embeddings
    from flair.config_parser import ConfigParser
    from flair.custom_data_loader import ColumnDataLoader
    from flair.data import Dictionary
    from flair.embeddings import StackedEmbeddings, TransformerWordEmbeddings
    from flair.list_data import ListCorpus
    from flair.models import FastSequenceTagger
    from flair.trainers import ModelFinetuner
    from flair.training_utils import Result
    return {"tagger": FastSequenceTagger, "trainer": ModelFinetuner, "corpus": ListCorpus, "config_parser": ConfigParser,
            "embedding": TransformerWordEmbeddings, "stacked_embeddings": StackedEmbeddings, "loader": ColumnDataLoader,
            "dictionary": Dictionary, "result": Result, "flair_module": flair, "datasets_module": flair.datasets,
            "embeddings_module": flair.embeddings}


def test_every_flair_import_of_train_py_resolves():
    for imp in SURFACE["imports"]:
        mod = importlib.import_module(imp["module"])
        if imp["name"] is not None:
            assert hasattr(mod, imp["name"]), "train.py:%d  from %s import %s" % (imp["line"], imp["module"], imp["name"])


def test_every_attribute_train_py_reads_exists():
    roles = _roles()
    missing = []
    for role, attrs in SURFACE["attributes"].items():
        if role == "params":
            from flair.utils.from_params import Params
            assert hasattr(Params, "__getitem__") and hasattr(Params, "from_file")
            continue
        target = roles[role]
        have = set(dir(target)) if inspect.ismodule(target) else _class_attrs(target)
        for name, uses in attrs.items():
            if not uses["load"]:
                continue  # train.py only assigns it (student.is_mst = True, embedding.v2_doc = True ...)
            if name.startswith("__"):
                assert hasattr(target, name), (role, name)
                continue
            if name not in have and (role, name) not in OUT_OF_SCOPE:
                missing.append("%s.%s (train.py:%s)" % (role, name, uses["load"][:3]))
    assert not missing, missing


def _callee(call, roles):
    owner, func = call["owner"], call["func"]
    if owner is None:
        from flair.trainers import ModelFinetuner, ReinforcementTrainer
        return {"ColumnDataLoader": [roles["loader"].__init__], "ConfigParser": [roles["config_parser"].__init__],
                "trainer_func": [ModelFinetuner.__init__, ReinforcementTrainer.__init__], "ListCorpus": [roles["corpus"].__init__]
                }.get(func)
    target = roles.get(owner)
    if target is None or (owner, func) in OUT_OF_SCOPE:
        return None
    fn = getattr(target, func, None)
    assert fn is not None, "train.py:%d calls %s.%s" % (call["line"], owner, func)
    if inspect.isclass(fn):
        fn = fn.__init__
    return [fn]


def test_every_keyword_train_py_passes_is_an_explicit_parameter():
    """**kwargs catch-alls do not count: a keyword only satisfies the surface if the callee names it"""
    roles = _roles()
    bad = []
    for call in SURFACE["calls"]:
        fns = _callee(call, roles)
        if not fns:
            continue
        for fn in fns:
            if fn in (torch_module_eval(), ):
                continue
            try:
                params = _explicit_params(fn)
            except (TypeError, ValueError):
                continue
            for kw in call["keywords"]:
                if kw == "professors" and fn.__qualname__.startswith("ReinforcementTrainer"):
                    continue   # train.py:127 is the distill_mode branch; the reference's ReinforcementTrainer has no such parameter either
                if kw not in params:
                    bad.append("train.py:%d %s(%s=...) not a parameter of %s" % (call["line"], call["func"], kw, fn.__qualname__))
    assert not bad, bad


def torch_module_eval():
    import torch
    return torch.nn.Module.eval


def test_corpus_list_keys_and_train_config_base_path():
    roles = _roles()
    p = _explicit_params(roles["corpus"].__init__)
    assert {"train", "dev", "test"} <= p           # ListCorpus(**{'train':[],'dev':[],'test':[]}), train.py:366-370
    assert "base_path" in _explicit_params(roles["trainer"].train)   # train_config['base_path'] = ..., train.py:135,412


@pytest.mark.parametrize("section", sorted(SURFACE["yaml_keywords"]))
def test_yaml_keyword_sets_are_explicit_parameters(section):
    """every key the shipped config/*.yaml files put in train: / <Trainer>: / model: / embeddings: is a named parameter"""
    import flair.embeddings as E
    from flair.models import FastSequenceTagger
    from flair.trainers import ModelFinetuner, ReinforcementTrainer
    keys = set(SURFACE["yaml_keywords"][section])
    kind, _, name = section.partition(":")
    if kind == "train":
        params = _explicit_params(ModelFinetuner.train) | _explicit_params(ReinforcementTrainer.train)
    elif kind == "trainer":
        params = _explicit_params({"ModelFinetuner": ModelFinetuner, "ReinforcementTrainer": ReinforcementTrainer}[name].__init__)
    elif kind == "model":
        assert name == "FastSequenceTagger"
        params = _explicit_params(FastSequenceTagger.__init__)
    else:
        cls = getattr(E, name, None)
        # (ELMoEmbeddings / FastWordEmbeddings need weight files that do not exist offline, SURVEY §8f-1: they construct as
        # deselected-only placeholders with the reference's keywords)
        assert cls is not None, name
        params = _explicit_params(cls.__init__)
    assert keys <= params, sorted(keys - params)


def test_every_shipped_yaml_block_is_accepted():
    """tests/golden/shipped_yaml_blocks.json = the model / train / trainer / embeddings blocks of all 16 YAMLs the reference ships
    (data extracted by oracle/gen_shipped_yaml_blocks.py).  Every key must be an explicit parameter of the mirror's constructor /
    train() (unknown keys would only be warned about), and no key may switch on a path the mirror refuses -- except in the one ACE
    YAML, which is inference-only here (ReinforcementTrainer.train raises; its ELMo / fastText embeddings need downloads)."""
    import inspect
    import json
    import flair.embeddings as E
    from flair.models.sequence_tagger_model import _UNSUPPORTED_TRUE, FastSequenceTagger
    from flair.trainers import ModelFinetuner, ReinforcementTrainer
    blocks = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "shipped_yaml_blocks.json")))
    assert len(blocks) == 16
    refused_train = ("use_amp", "use_autocast", "rootschedule", "freezing", "use_unlabeled_data", "unlabeled_data_for_zeroshot",
                     "gold_reward", "language_attention_warmup", "language_attention_warmup_and_fix")
    n_finetune = n_multiview = 0
    for name, b in blocks.items():
        ace = b["trainer"] == "ReinforcementTrainer"
        cls = {"ModelFinetuner": ModelFinetuner, "ReinforcementTrainer": ReinforcementTrainer}[b["trainer"]]
        init_p = set(inspect.signature(cls.__init__).parameters)
        assert set(b["trainer_block"]) <= init_p, (name, sorted(set(b["trainer_block"]) - init_p))
        assert not any(b["trainer_block"].get(k) for k in ("distill_mode", "ensemble_distill_mode", "train_with_professor")), name
        (mcls, mkw), = b["model"].items()
        assert mcls == "FastSequenceTagger"
        mp = set(inspect.signature(FastSequenceTagger.__init__).parameters)
        assert set(mkw) <= mp, (name, sorted(set(mkw) - mp))
        assert not [k for k in mkw if k in _UNSUPPORTED_TRUE and mkw[k]], name
        if mkw.get("multi_view_training"):
            assert mkw.get("distill_posterior") and mkw.get("remove_x") and not mkw.get("use_rnn"), name   # the implemented form
            n_multiview += 1
        for ekey, ekw in b["embeddings"].items():
            ecls = getattr(E, ekey.split("-")[0], None)
            assert ecls is not None, (name, ekey)
            ep = set(inspect.signature(ecls.__init__).parameters)
            assert set(ekw or {}) <= ep, (name, ekey, sorted(set(ekw or {}) - ep))
        if ace:
            continue
        n_finetune += 1
        tp = set(inspect.signature(ModelFinetuner.train).parameters)
        assert set(b["train"]) <= tp, (name, sorted(set(b["train"]) - tp))
        assert not [k for k in refused_train if b["train"].get(k)], name
        assert list(b["embeddings"]) == ["TransformerWordEmbeddings-0"] and not mkw.get("use_rnn"), name
    assert n_finetune == 15 and n_multiview == 3


def test_reference_entry_script_form(tmp_path):
    """the reference's own train.py is run through `python -m kbner.run_script <script>`.  A stand-in script is
    generated from the fixture (every import statement the reference's train.py makes, in order) and placed next to a DECOY
    `flair/` package -- the layout of the reference checkout, where the script's directory would come first on sys.path.
    `python <script>` picks the decoy (why the plain form cannot work); the run_script form imports this repo's mirror."""
    import subprocess
    import sys
    repo = os.path.dirname(HERE)
    pkg = os.path.join(repo, "kb-ner_amd")
    lines = []
    for imp in SURFACE["imports"]:
        if imp["name"] is None:
            lines.append("import %s%s" % (imp["module"], (" as " + imp["asname"]) if imp.get("asname") else ""))
        elif (("datasets_module", imp["name"]) in OUT_OF_SCOPE) or imp["name"] == "*":
            continue
        else:
            lines.append("from %s import %s" % (imp["module"], imp["name"]))
    lines += ["import sys, flair", "print('FLAIR_FROM', flair.__file__)", "print('ARGV', sys.argv[1:])"]
    script = tmp_path / "train.py"
    script.write_text("\n".join(lines) + "\n")
    decoy = tmp_path / "flair"
    decoy.mkdir()
    (decoy / "__init__.py").write_text("print('DECOY_FLAIR_IMPORTED')\nraise ImportError('decoy: the reference package shadows PYTHONPATH')\n")
    env = dict(os.environ, PYTHONPATH=pkg)
    plain = subprocess.run([sys.executable, str(script), "--config", "x.yaml"], env=env, cwd=str(tmp_path), capture_output=True, text=True)
    assert plain.returncode != 0 and "DECOY_FLAIR_IMPORTED" in plain.stdout, (plain.stdout, plain.stderr)
    good = subprocess.run([sys.executable, "-m", "kbner.run_script", str(script), "--config", "x.yaml"], env=env, cwd=str(tmp_path),
                          capture_output=True, text=True)
    assert good.returncode == 0, (good.stdout, good.stderr)
    assert "DECOY_FLAIR_IMPORTED" not in good.stdout
    assert ("FLAIR_FROM " + os.path.join(pkg, "flair")) in good.stdout, good.stdout
    assert "ARGV ['--config', 'x.yaml']" in good.stdout
