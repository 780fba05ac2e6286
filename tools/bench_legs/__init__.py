"""The legs of bench.py (one module per kind of measurement); bench.py itself is the driver contract + the orchestration."""
