"""bench.py's legs, one module per leg, imported when the leg runs (bench.py itself holds the argument parsing, the rank launch and the
order of the legs; `line.py` builds the one compact stdout line)."""
