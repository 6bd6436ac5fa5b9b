"""efficientconformer_amd — MI355X (gfx950) native Efficient Conformer encoder forward path.

Host-side mirror of the reference's encoder interface (models/encoders.ConformerEncoder,
models/model_ctc.ModelCTC) over the C ABI of include/effconf.h; the compute is hand-written HIP in
csrc/.  See DESIGN.md / INTEGRATION.md.
"""
from .config import build_plan, load_config, named_config  # noqa: F401
from .encoders import ConformerEncoder  # noqa: F401
from .model_ctc import ModelCTC  # noqa: F401
from .transducer import Transducer  # noqa: F401
from .checkpoint import load_checkpoint, save_checkpoint  # noqa: F401
from .batching import FrontDoor, bucket_batches, collate_fn_pad  # noqa: F401

__all__ = ["ConformerEncoder", "ModelCTC", "Transducer", "load_checkpoint", "save_checkpoint", "FrontDoor", "bucket_batches", "collate_fn_pad", "build_plan", "load_config", "named_config"]
