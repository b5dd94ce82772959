"""Instance-level validation metrics on the MI355X (reference: skoots/validate/lib.py)."""
from .lib import accuracies_from_iou, f1_score, get_segmentation_errors, mask_iou  # noqa: F401
