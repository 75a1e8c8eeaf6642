"""Loss-weight generators of the label-history branch (decaying only; see tracking_methods.py)."""
