import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from scenes import random_gaussians
from hip_helpers import hip_forward
for P, W, H, kw in [(20000, 160, 96, {}), (60000, 320, 192, {}), (30000, 160, 96, dict(scale=(-3.8, -1.8))), (100000, 480, 270, {})]:
    a = hip_forward(random_gaussians(P, seed=52, W=W, H=H, **kw))
    n = (a["ranges"][:, 1] - a["ranges"][:, 0]).astype(np.int64)
    print(P, W, H, kw, "min", n[n > 0].min(), "median", int(np.median(n)), "max", n.max(), "tiles>1536:", int((n > 1536).sum()), "of", len(n))
