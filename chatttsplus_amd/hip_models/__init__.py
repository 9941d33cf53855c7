"""`infer_type: "hip"` model classes: same names / call surface as chattts_plus.models (reference
chattts_plus/models/__init__.py) so the pipeline's getattr(models, name)(**kwargs) dispatch
(pipelines/chattts_plus_pipeline.py:113-129) can select them."""
from .gpt import GPT  # noqa: F401
from .vocoder import DVAE, Synth, Vocos  # noqa: F401
from .encoder import DVAEEncoder  # noqa: F401
