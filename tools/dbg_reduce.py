import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multimotionfusion_amd import cudafuncs as cf
ctx = cf.Context(0)
for (w, h) in ((64, 48), (160, 120), (640, 480)):
    img = np.full((h, w), 100, np.uint8)
    img[:, ::2] = 50
    d = torch.from_numpy(img).cuda()
    I = np.eye(3, dtype=np.float32)
    A, b, res = cf.so3Step(ctx, d, d, I, I, I)
    print(w, h, "so3 count", res, "expected", (w - 2) * (h - 2), "A00", A[0, 0])
