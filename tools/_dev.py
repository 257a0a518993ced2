"""Developer tools: TOOLS_DEV=1 in the environment makes a tool load libvidmem_dev.so (make -C csrc dev), where the
VIDMEM_* developer switches are read; without it the release library runs."""
import os


def maybe_dev():
    if os.environ.get("TOOLS_LIB"):      # any other build of the library (an older commit's, for a bisection)
        from vidmem import _lib
        _lib.LIB_PATH = os.path.abspath(os.environ["TOOLS_LIB"])
    elif os.environ.get("TOOLS_DEV"):
        from vidmem import _lib
        _lib.use_dev_library()
