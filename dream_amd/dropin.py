"""Run an UNMODIFIED NVlabs/DREAM checkout on dream_amd -- no edit to the reference's files.

The reference's scripts reach the hot path through ``import dream`` (/root/reference/dream/__init__.py:1-9 star-imports
``network``, ``models``, ``image_proc`` ...; scripts/train_network.py:19,403-408,505,570; dream/analysis.py:17,147,210).
This module rebinds the hot-path names inside that package at import time:

  * ``dream.network``, ``dream.models``, ``dream.spatial_softmax``  ->  the dream_amd modules of the same name (entered into
    ``sys.modules`` before ``dream`` is imported, so ``from .network import *`` in the reference's ``__init__`` star-imports
    dream_amd's ``DreamNetwork``, ``create_network_from_config_file/_data``, ``KNOWN_*``, ``DreamHourglass``, ``ResnetSimple``;
    the reference's own files of those names are never executed);
  * ``dream.image_proc`` stays the reference's module (visualisation, PIL preprocessing ...) except
    ``peaks_from_belief_maps`` (dream/image_proc.py:914-1018), which is re-pointed at the HIP peak stage after the module has
    been loaded -- in ``dream.image_proc`` and in the ``dream`` namespace the star import copied it into.

Use either of

    python -m dream_amd.dropin scripts/train_network.py -i DATA -m manip.yaml -ar arch.yaml ...     (from the DREAM checkout)
    import dream_amd.dropin        # anywhere before the first ``import dream`` (sitecustomize, a launcher, a notebook cell)

``install()`` is idempotent; ``uninstall()`` removes the hooks (it cannot un-import an already imported ``dream``).
"""
import importlib.abc
import importlib.util
import runpy
import sys

_REBOUND = ("network", "models", "spatial_softmax")
_installed = [None]


def _rebind_package(pkg):
    """After the reference's dream/__init__ has run: make attribute access (``dream.network.X``) agree with sys.modules and
    re-point the peak stage the star import copied into the package namespace."""
    import dream_amd
    # dream_amd.network's public names include the dream_amd modules it imports (image_proc, models, ops): the reference's
    # star import copied those over the package's own submodule attributes -- put every submodule back where it belongs
    for full, mod in list(sys.modules.items()):
        if full.startswith(pkg.__name__ + ".") and "." not in full[len(pkg.__name__) + 1:] and mod is not None:
            setattr(pkg, full[len(pkg.__name__) + 1:], mod)
    for name in _REBOUND:
        setattr(pkg, name, getattr(dream_amd, name))
    for name in ("DreamNetwork", "create_network_from_config_file", "create_network_from_config_data", "KNOWN_ARCHITECTURES",
                 "KNOWN_OPTIMIZERS", "DreamHourglass", "DreamHourglassMultiStage", "ResnetSimple"):
        src = dream_amd.network if hasattr(dream_amd.network, name) else dream_amd.models
        setattr(pkg, name, getattr(src, name))
    pkg.peaks_from_belief_maps = dream_amd.image_proc.peaks_from_belief_maps
    pkg.__dream_amd_dropin__ = dream_amd.__version__


def _rebind_image_proc(mod):
    import dream_amd
    mod.peaks_from_belief_maps = dream_amd.image_proc.peaks_from_belief_maps


_HOOKS = {"dream": _rebind_package, "dream.image_proc": _rebind_image_proc}


class _PostImportFinder(importlib.abc.MetaPathFinder):
    """Lets the regular finders locate ``dream`` / ``dream.image_proc`` and runs a hook once their module body has executed."""

    def find_spec(self, fullname, path=None, target=None):
        if fullname not in _HOOKS:
            return None
        for finder in sys.meta_path:
            if finder is self or not hasattr(finder, "find_spec"):
                continue
            spec = finder.find_spec(fullname, path, target)
            if spec is not None:
                break
        else:
            return None
        if spec.loader is None or not hasattr(spec.loader, "exec_module"):
            return spec
        spec.loader = _PostImportLoader(spec.loader, _HOOKS[fullname])
        return spec


class _PostImportLoader(importlib.abc.Loader):
    def __init__(self, inner, hook):
        self.inner, self.hook = inner, hook

    def create_module(self, spec):
        return self.inner.create_module(spec)

    def exec_module(self, module):
        self.inner.exec_module(module)
        self.hook(module)

    def __getattr__(self, name):                    # get_filename, get_source, is_package ... of the wrapped loader
        return getattr(self.inner, name)


def install():
    """Idempotent.  Must run before the first ``import dream`` to keep the reference's network.py / models.py from executing;
    called later, it still rebinds the names in an already imported package."""
    import dream_amd
    if _installed[0] is None:
        finder = _PostImportFinder()
        sys.meta_path.insert(0, finder)
        _installed[0] = finder
    for name in _REBOUND:
        sys.modules["dream." + name] = getattr(dream_amd, name)
    if "dream.image_proc" in sys.modules:
        _rebind_image_proc(sys.modules["dream.image_proc"])
    if "dream" in sys.modules and getattr(sys.modules["dream"], "__dream_amd_dropin__", None) is None:
        _rebind_package(sys.modules["dream"])
    return dream_amd


def uninstall():
    if _installed[0] is not None:
        if _installed[0] in sys.meta_path:
            sys.meta_path.remove(_installed[0])
        _installed[0] = None
    import dream_amd
    for name in _REBOUND:
        if sys.modules.get("dream." + name) is getattr(dream_amd, name):
            del sys.modules["dream." + name]


install()


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        raise SystemExit("usage: python -m dream_amd.dropin <script.py> [script arguments ...]")
    install()
    sys.argv = argv
    runpy.run_path(argv[0], run_name="__main__")


if __name__ == "__main__":
    main()
