"""trinerflet_amd -- MI355X-native hot path of TriNeRFLet (wavelet-triplane NeRF).

Sub-packages mirror the reference's module names:
  raymarching, shencoder, triplaneencoder.triplane_encoder, encoding, activation,
  nerf.network, nerf.renderer
`install_dropin()` registers them under the reference's top-level names so that
reconstruction/main_nerf.py imports them unchanged.
"""
import importlib
import sys

__version__ = "0.1.0"

_DROPIN = ("raymarching", "shencoder", "triplaneencoder", "encoding", "activation", "nerf")


def install_dropin():
    """Alias trinerflet_amd.<name> as top-level <name> (what main_nerf.py:15-16 expects on sys.path)."""
    for name in _DROPIN:
        sys.modules[name] = importlib.import_module(f"trinerflet_amd.{name}")
    sys.modules["triplaneencoder.triplane_encoder"] = importlib.import_module(
        "trinerflet_amd.triplaneencoder.triplane_encoder")
    sys.modules["nerf.network"] = importlib.import_module("trinerflet_amd.nerf.network")
    sys.modules["nerf.renderer"] = importlib.import_module("trinerflet_amd.nerf.renderer")
