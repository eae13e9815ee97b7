import sys, os
sys.path.insert(0, os.getcwd())
import torch
from graph_amd import synth
for scale in (22, 24, 26):
    src, dst = synth.rmat_edges(scale, 42)
    key = (src.long() << 32) | dst.long()
    del src, dst
    key, _ = torch.sort(key)
    distinct = int((key[1:] != key[:-1]).sum()) + 1
    m = key.numel()
    # multiplicity histogram of runs
    starts = torch.nonzero(torch.cat([torch.ones(1, dtype=torch.bool, device=key.device), key[1:] != key[:-1]])).flatten()
    lens = torch.diff(torch.cat([starts, torch.tensor([m], device=key.device)]))
    capped = int(((lens + 2) // 3).sum())  # entries if multiplicity is capped at 3
    print(scale, m, distinct, round(distinct / m, 4), 'capped3 entries', round(capped / m, 4), 'max mult', int(lens.max()))
    del key, starts, lens
    torch.cuda.empty_cache()
