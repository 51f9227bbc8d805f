"""One process that runs the PLDA self-score of 4874 x-vectors 30 times: the subject of a rocprofv3 --kernel-trace --stats run."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "oracle"))
import numpy as np, torch
from xvector_amd import scoring
import plda_oracle as po
dim = 512
mean, F, Sigma = po.make_plda(dim, 200, seed=21)
scorer = scoring.PldaScorer(mean, F, Sigma)
x = torch.randn((4874, dim), device="cuda:0", dtype=torch.float64) + torch.from_numpy(mean).to("cuda:0")
for _ in range(30): scorer.score(x)
torch.cuda.synchronize()
