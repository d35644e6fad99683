"""hydra stand-in for the golden-vector harness only."""
