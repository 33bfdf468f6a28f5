"""trinerflet_amd -- MI355X-native hot path of TriNeRFLet (wavelet-triplane NeRF).

Sub-packages mirror the reference's module names:
  raymarching, shencoder, triplaneencoder.triplane_encoder, encoding, activation,
  nerf.network, nerf.renderer
`install_dropin()` registers them under the reference's top-level names so that
reconstruction/main_nerf.py imports them unchanged; `install_backends()` only puts the native-name
modules `_raymarching` / `_shencoder` on sys.path (the names aux_libs/*/ bind with `import _x as _backend`).
"""
import importlib
import os
import sys

__version__ = "0.2.0"

_TOP = ("raymarching", "shencoder", "triplaneencoder", "encoding", "activation", "ffmlp")
_BACKENDS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "backends")


def install_backends():
    """Make `import _raymarching` / `import _shencoder` (aux_libs/raymarching/raymarching.py:9-12,
    aux_libs/shencoder/sphere_harmonics.py:9-12) resolve to the ctypes-backed modules of trinerflet_amd/backends."""
    if _BACKENDS not in sys.path:
        sys.path.insert(0, _BACKENDS)


def install_dropin(windowed_autograd=True, fused_adam=False):
    """Register the MI355X build under the reference's top-level names (main_nerf.py:15-16 puts aux_libs/ on sys.path
    and reconstruction/ is the script directory):

      raymarching, shencoder, triplaneencoder[.triplane_encoder], encoding, activation, ffmlp  -> trinerflet_amd.<name>
      nerf.network, nerf.renderer, nerf.network_ff (the `--ff` switch)                        -> trinerflet_amd.nerf.<name>

    The reference's `nerf` package itself is NOT replaced: `nerf.provider`, `nerf.utils` (its Trainer), ... keep
    coming from reconstruction/nerf/ -- only the two hot-path modules inside it are overridden.  When no `nerf` package
    is importable (stand-alone use), trinerflet_amd.nerf takes the name.

    windowed_autograd: encoders constructed afterwards rebuild / differentiate only the occupancy window of the planes
    in a training iteration's get_planes() (TriPlaneVolume._autograd_window: enough for the reference's Trainer, whose
    loop discards get_planes()'s result and renders marched samples; 8.4 -> 7.1 ms per step at the base configuration).
    Pass False for code that reads whole planes under autograd.

    fused_adam: `torch.optim.Adam(...)` over dense fp32 device parameters returns trinerflet_amd.optim.FusedAdamL1
    (optim.patch_torch_adam: one pass per parameter, the regulariser's gradient and GradScaler's work folded in; same
    state_dict layout), so that main_nerf.py:119 needs no edit either: 14.5 -> 7.1 ms per step at the base configuration.
    Off by default -- it replaces a name in torch's namespace for the whole process."""
    install_backends()
    importlib.import_module("trinerflet_amd.triplaneencoder.triplane_encoder").WINDOWED_AUTOGRAD = bool(windowed_autograd)
    optim = importlib.import_module("trinerflet_amd.optim")
    (optim.patch_torch_adam if fused_adam else optim.unpatch_torch_adam)()
    for name in _TOP:
        sys.modules[name] = importlib.import_module(f"trinerflet_amd.{name}")
    sys.modules["triplaneencoder.triplane_encoder"] = importlib.import_module(
        "trinerflet_amd.triplaneencoder.triplane_encoder")
    ours = importlib.import_module("trinerflet_amd.nerf")
    pkg = sys.modules.get("nerf")
    if pkg is None:
        try:
            pkg = importlib.import_module("nerf")      # the reference's (a namespace package: no __init__.py)
        except ImportError:
            pkg = None
    if pkg is None or not hasattr(pkg, "__path__"):
        pkg = sys.modules["nerf"] = ours
    for sub in ("network", "renderer", "network_ff"):
        mod = importlib.import_module(f"trinerflet_amd.nerf.{sub}")
        sys.modules[f"nerf.{sub}"] = mod
        setattr(pkg, sub, mod)
