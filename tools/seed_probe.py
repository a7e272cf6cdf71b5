#!/usr/bin/env python3
"""Would cross-instance neighbour seeding (SURVEY 8f next-4 / VERDICT r2 item 5) shorten the cold first second
of the headline sweep?  CPU-only probe with the oracle (bounded 16-entry store, sample by sample): the samples on
which a grid cell has to solve the hard way (base solve > 5 iterations: the ones the CachingSolver stores) for a
cell, its wave-mates (level pot differs), a block-mate, the cell of the next block (tone differs) and of the next
drive value.  Result (round 3): 239 of 247 hard samples of a wave-mate COINCIDE with the cell's own -- the
instances of a block are driven by the same input in lockstep, so when one of them needs a start point its
neighbours are solving the very same hard point in the same wave pass and have nothing to offer yet; a stored
solution only pays off one signal period later, and by then every instance holds its own.  Seeding from
block-mates therefore cannot reduce the cold cost of this workload."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, bench
from helpers import load, sine
from oracle.refpy import RefRunner
from acme_jl_amd.model import CachingHomotopySolver
T=4410*3
_, pots, amp = bench.grid_inputs("superover_grid", 0, 1, 8192, T)
m = load("superover_var", CachingHomotopySolver)
def trace(i):
    r = RefRunner(m); r.set_cache_limit(16)
    u = np.zeros((4,T)); u[0]=amp*sine(T); u[1:]=pots[i][:,None]
    its=np.zeros(T,int)
    for n in range(T):
        r.run(u[:,n:n+1]); its[n]=r.report.iters_total
    return its
base=5000
cells=[base, base+1, base+3, base+15, base+16, base+256]   # wave-mates (level), block-mates, next block (tone), next drive
print([tuple(np.round(pots[c],3)) for c in cells])
tr={c:trace(c) for c in cells}
hard={c:set(np.nonzero(tr[c]>5)[0]) for c in cells}
for c in cells:
    h=sorted(hard[c])
    print(c, 'iters/sample %.2f'%tr[c].mean(), 'stored', len(h), 'first', h[:12])
a=hard[base]
for c in cells[1:]:
    b=hard[c]
    # for each hard sample of c: was there a hard sample of base at an EARLIER time within 40 samples (could have seeded)?
    earlier=sum(1 for n in b if any((n-d) in a for d in range(1,40)))
    same=len(a&b)
    print(f"cell {c} vs {base}: hard samples {len(b)}, coincide exactly {same}, donor finished earlier (1..39 samples) {earlier}")
