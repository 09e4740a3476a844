"""GPU-side helpers for the parity tests (torch is used only for device memory/streams)."""
import numpy as np
import torch

import vfgs_testlib as T


class DevFrame:
    """A Frame (tests/vfgs_testlib.py) resident on cuda:0, same geometry, planes contiguous."""

    def __init__(self, f: T.Frame):
        self.f = f
        self.Y = torch.from_numpy(f.Y.view(np.uint8).copy()).cuda()
        self.U = torch.from_numpy(f.U.view(np.uint8).copy()).cuda()
        self.V = torch.from_numpy(f.V.view(np.uint8).copy()).cuda()

    def ptrs(self, y=0):
        f = self.f
        sz = f.Y.itemsize
        return (self.Y.data_ptr() + y * f.stride * sz,
                self.U.data_ptr() + (y // f.suby) * f.cstride * sz,
                self.V.data_ptr() + (y // f.suby) * f.cstride * sz)

    def download(self) -> T.Frame:
        torch.cuda.synchronize()
        g = self.f.copy()
        g.Y[...] = self.Y.cpu().numpy().view(g.dtype)
        g.U[...] = self.U.cpu().numpy().view(g.dtype)
        g.V[...] = self.V.cpu().numpy().view(g.dtype)
        return g


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream
