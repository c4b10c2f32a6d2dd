"""Import-path alias of merv_amd.load (the reference's merv/models/load_vid.py)."""
from merv_amd.load import available_model_names, available_models, get_model_description, load_vid  # noqa: F401
