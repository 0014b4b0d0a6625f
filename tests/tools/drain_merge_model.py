import numpy as np, sys
steps = np.load("/tmp/steps.npy")
n = steps.size
W = int(sys.argv[1]) if len(sys.argv) > 1 else 7168     # waves (7 WG/CU)
REFILL = 8
rng = np.random.default_rng(2)

def wave_run(seq, K):
    """returns (trips until supply exhausted and active <= K, residuals then, total trips without merge)"""
    rem = np.zeros(64, dtype=np.int64)
    cur = 0; trips = 0
    ready_trips = None; ready_rem = None
    while True:
        idle = rem == 0
        ni = int(idle.sum())
        if cur < seq.size and ni >= REFILL:
            take = min(ni, seq.size - cur)
            idx = np.flatnonzero(idle)[:take]
            rem[idx] = seq[cur:cur + take]; cur += take
        act = rem > 0
        na = int(act.sum())
        if cur >= seq.size and ready_trips is None and na <= K:
            ready_trips = trips; ready_rem = rem[act].copy()
        if na == 0:
            if cur >= seq.size: break
            continue
        k = int(rem[act].min())
        rem[act] -= k; trips += k
    return ready_trips, ready_rem, trips

def model(batches_per_wave, K, nwg=200):
    base = []; merged = []
    nb = n // 64
    for g0 in rng.choice(W // 4, nwg, replace=False):
        tot0 = 0; tot1 = 0; res = []
        for w in range(4 * g0, 4 * g0 + 4):
            ids = []
            for b in range(batches_per_wave):
                gb = (b * W + w) % nb          # (wrap: model longer supplies with the same ray population)
                ids.append(np.arange(gb * 64, gb * 64 + 64))
            seq = steps[np.concatenate(ids)]
            rt, rr, t = wave_run(seq, K)
            tot0 += t; tot1 += rt; res.append(rr)
        r = np.concatenate(res)
        tot1 += int(r.max()) if r.size else 0
        base.append(tot0); merged.append(tot1)
    return np.mean(base), np.mean(merged)

for bpw, label in ((4, "isolated 2.07 M-ray launch (4 batches per wave)"), (13, "fused batch chunk (13 batches per wave)"), (34, "34 batches per wave")):
    ideal = steps.mean() * bpw * 4          # trips per WG at full lanes
    for K in (16, 32):
        b, m = model(bpw, K)
        print(f"{label}: K={K}: WG trips no-merge {b:.0f} merge {m:.0f} ({100*(b-m)/b:.1f} % fewer); ideal {ideal:.0f}")
