"""The one helper of reconstruction/triplaneencoder/utils.py that is on the hot path."""
import math


def get_levels(upscale_factor):
    # reference: triplaneencoder/utils.py:274-279
    wavelet_levels = math.log2(upscale_factor)
    if abs(wavelet_levels - round(wavelet_levels)) > 1e-5:
        raise ValueError('Unsupported res. should be 2^')
    return round(wavelet_levels)
