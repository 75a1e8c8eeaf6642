"""TEST INFRASTRUCTURE ONLY — CPU oracle of the hot path (see model_oracle.py / path_oracle.py).
Never imported by the product package; only by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py."""
