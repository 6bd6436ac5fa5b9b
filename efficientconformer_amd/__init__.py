"""efficientconformer_amd — MI355X (gfx950) native Efficient Conformer encoder forward path.

Host-side mirror of the reference's encoder interface (models/encoders.ConformerEncoder,
models/model_ctc.ModelCTC) over the C ABI of include/effconf.h; the compute is hand-written HIP in
csrc/.  See DESIGN.md / INTEGRATION.md.
"""
from .config import build_plan, load_config, named_config  # noqa: F401
from .encoders import ConformerEncoder  # noqa: F401
from .model_ctc import ModelCTC  # noqa: F401
from .transducer import Transducer  # noqa: F401

__all__ = ["ConformerEncoder", "ModelCTC", "Transducer", "build_plan", "load_config", "named_config"]
