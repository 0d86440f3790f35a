"""The by-name weight recipe / synthetic inputs live in clc_amd/recipe.py (a neutral, torch-only module shared by the
benchmark, the product tests and this checker); re-exported here for the checker-side callers."""
from clc_amd.recipe import CONV_GAIN, apply_weight_recipe, synthetic_image  # noqa: F401
