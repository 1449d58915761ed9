"""Worker of tests/test_gpu_flair_e2e.py::test_trainer_two_ranks_one_gpu_equals_accumulation (not a test module): one
ModelFinetuner.train run on the tiny end-to-end corpus, as rank r of WORLD_SIZE processes that share ONE GPU (gloo collectives on
device tensors) or as a single process; rank 0 writes the final parameters and histories to <out>.
usage: python dp_trainer_gpu_worker.py <cfg.yaml> <out.pt> <gradient_accumulation_steps>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kb-ner_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

W = int(os.environ.get("WORLD_SIZE", "1"))
if W > 1:   # before `import flair` (which would pick RCCL on a GPU box: two ranks cannot share a device there)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    os.environ["LOCAL_RANK"] = "0"      # flair.device = cuda:LOCAL_RANK: both ranks of this test live on device 0
torch.cuda.set_device(0)
import yaml  # noqa: E402
from flair.config_parser import ConfigParser  # noqa: E402
from flair.trainers import ModelFinetuner  # noqa: E402
from flair.utils.from_params import Params  # noqa: E402

cfg_path, out_path, accum = sys.argv[1], sys.argv[2], int(sys.argv[3])
cfg = yaml.safe_load(open(cfg_path))
cfg["train"]["gradient_accumulation_steps"] = accum
cfg["train"]["fuse_accumulation"] = False   # the single process runs the two micro-batches of a step as the two ranks do: one by one
cfg["target_dir"] = os.path.join(os.path.dirname(out_path), "out_w%d" % W)
mine = os.path.join(os.path.dirname(out_path), "cfg_w%d_r%s.yaml" % (W, os.environ.get("RANK", "0")))
with open(mine, "w") as f:
    yaml.safe_dump(cfg, f)
torch.manual_seed(11)
from kbner import engine as _engine  # noqa: E402
_norms = []
_step = _engine.FusedAdamW.step


def _recording_step(self, grad_scale=1.0):
    out = _step(self, grad_scale=grad_scale)
    _norms.append(float(out) ** 0.5 * grad_scale)      # the clip norm of the MEAN gradient (device sync: test only)
    return out


_engine.FusedAdamW.step = _recording_step
cp = ConfigParser(Params.from_file(mine))
student = cp.create_student()
trainer = ModelFinetuner(student, None, cp.corpus, config=cp.config, **cp.config["ModelFinetuner"])
trainer.lazy_embedding_rows = "always"     # the tiny vocabulary would otherwise keep the eager row update
out = trainer.train(cp.get_target_path, **cp.config["train"])
opt = trainer.optimizer
assert opt.lazy_rows, "lazy embedding rows were requested"
opt.materialize()
torch.cuda.synchronize()
if not dist.is_initialized() or dist.get_rank() == 0:
    a = student.engine.arena
    torch.save({"p": a.p.detach().cpu(), "m": a.m.detach().cpu(), "train_loss_history": out["train_loss_history"],
                "dev_score_history": out["dev_score_history"], "t": opt.t, "world": W,
                "live_rows": int((a.emb_flags != 0).sum()),
                "clip_norms": _norms, "offsets": dict(a.offsets), "shapes": dict(a.shapes)}, out_path)
if dist.is_initialized():
    dist.barrier()
    dist.destroy_process_group()
