"""Import-name alias so that scripts written against the reference package keep working unchanged:
`from merv import load_vid` (scripts/quick_start.py:5 of the reference), `merv.available_models()`, ... resolve to the
MI355X build in `merv_amd`. Nothing is implemented here."""
from merv_amd.load import available_model_names, available_models, get_model_description, load_vid  # noqa: F401

__all__ = ["available_models", "available_model_names", "get_model_description", "load_vid"]
