"""Stand-in for `simple_knn._C` (reference submodules/simple-knn/ext.cpp:14-16): distCUDA2 on libmom4d."""
import torch

from .. import _native as N


def distCUDA2(points):
    """float32 [P,3] on the GPU -> float32 [P]: mean squared distance to the 3 nearest neighbours
    (simple-knn/spatial.cu:15-25)."""
    if not points.is_cuda:
        raise N.MomError("distCUDA2: libmom4d has no CPU path")
    lib = N.lib()
    pts = points.contiguous().float()
    P = pts.shape[0]
    out = torch.zeros((P,), dtype=torch.float32, device=pts.device)
    if P:
        scratch = torch.empty((lib.mom_knn_scratch_bytes(P),), dtype=torch.uint8, device=pts.device)
        N.check(lib.mom_knn_mean_dist2(P, pts.data_ptr(), out.data_ptr(), scratch.data_ptr(), N.current_stream()),
                "mom_knn_mean_dist2")
    return out
