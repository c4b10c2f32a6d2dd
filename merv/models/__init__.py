"""Import-path alias: `from merv.models.load_vid import load_vid` (scripts/eval_*.py of the reference)."""
from merv_amd.load import available_model_names, available_models, get_model_description, load_vid  # noqa: F401
