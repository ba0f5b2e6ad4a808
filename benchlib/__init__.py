"""The legs of bench.py (the driver stays at the repo root: `python bench.py --gpus N --steps K --warmup W`)."""
