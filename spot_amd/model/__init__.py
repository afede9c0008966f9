"""Module names of the reference's `model` package (model/shift.py, model/modules.py,
model/common.py, model/min_gru.py), re-implemented as parameter containers whose arithmetic
runs in the HIP engine (spot_amd.engine)."""
