"""Workload of the counter passes for the next rows N3 / N4 (profiles/run_round.sh): the MFCC kernel on 256 x 3 s waveforms and
the PLDA score matrix of 4874 x-vectors, a few launches each (no oracle, no timing)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import xvector_amd as xa
dev = torch.device("cuda:0")
fe = xa.MfccFrontEnd(device=dev)
w = 0.1 * torch.randn(256, 48000, device=dev)
for _ in range(10):
    fe(w)
mean, F, Sigma = xa.synth.make_plda(512, 200, seed=21)
sc = xa.scoring.PldaScorer(mean, F, Sigma, device=dev)
x = torch.randn(4874, 512, device=dev).double() + torch.from_numpy(mean).to(dev)
for _ in range(4):
    sc.score(x)
torch.cuda.synchronize()
