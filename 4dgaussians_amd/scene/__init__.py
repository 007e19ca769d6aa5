"""MI355X counterparts of the reference's scene/hexplane.py and scene/deformation.py (the
deformation field only; the rest of the reference's `scene` package is out of scope)."""
