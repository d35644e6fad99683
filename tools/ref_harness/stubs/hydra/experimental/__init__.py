def initialize(*a, **k):
    raise RuntimeError("hydra stub: initialize() unavailable")


def compose(*a, **k):
    raise RuntimeError("hydra stub: compose() unavailable")
