"""chimera-st_amd — MI355X-native (gfx950) hot path of Chimera-ST behind the reference's fairseq
registry surface.  The directory name is hyphenated as the brief names it; import it with
`importlib.import_module("chimera-st_amd")` (a hyphen is only illegal in the `import` statement)."""
from . import lib  # noqa: F401  (ctypes binding; loading the .so is deferred to first use)

__all__ = ["lib"]
