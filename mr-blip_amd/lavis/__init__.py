"""Host-side mirror of the reference's LAVIS surface for the Mr. BLIP path ONLY (registry / Config / blip2_mr model /
moment_retrieval task / runner_base / datasets needed by train.py and evaluate.py).  Written from scratch; the compute is
mrblip.engine (HIP kernels).  Mirrors lavis/__init__.py:10-31 in what it registers: the library root path."""
import os

from lavis.common.registry import registry

root_dir = os.path.dirname(os.path.abspath(__file__))
registry.register_path("library_root", root_dir)
registry.register_path("repo_root", os.path.join(root_dir, ".."))
registry.register("MAX_INT", 1 << 31)
registry.register("SPLIT_NAMES", ["train", "val", "test"])

from lavis import models, tasks, runners, datasets  # noqa: E402,F401  (registration side effects)
from lavis.common import optims  # noqa: E402,F401
