"""Helpers for the -m gpu parity tests: they drive the product through its public surfaces (the
nn.Module drop-ins and the engine wrappers over the C ABI) and compare with the CPU oracle."""
import importlib

import torch

PKG = 'single-image-super-resolution_amd'


def pkg(sub=None):
    return importlib.import_module(PKG + ('.' + sub if sub else ''))


class FakeConv:
    """minimal ConvRef stand-in for kernel-level tests"""

    def __init__(self, weight, bias, geom, u=None, v=None):
        self.weight, self.bias, self.geom, self.u, self.v = weight, bias, geom, u, v


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def maxrel(a, b):
    a = a.detach().double().cpu().reshape(-1)
    b = b.detach().double().cpu().reshape(-1)
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))
