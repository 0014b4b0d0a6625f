"""Developer probe: CPU model of the persistent-wave scheduler of k_extend6 on the oracle's real
per-ray step counts (/tmp/steps.npy from orc.extend_steps).  One trip = one step for every active
lane; idle lanes refill when >= REFILL lanes are idle.  Prints trips per wave for a few policies."""
import sys
import numpy as np

steps = np.load(sys.argv[1] if len(sys.argv) > 1 else "/tmp/steps.npy").astype(np.int64)
n = steps.size
W = 8192
rng = np.random.default_rng(1)


def wave_trips(seq, refill=16, lanes=64):
    """seq: step counts of the wave's rays in processing order"""
    rem = np.zeros(lanes, dtype=np.int64)
    cur = 0
    trips = 0
    useful = 0
    while True:
        idle = rem == 0
        ni = int(idle.sum())
        if cur < seq.size and ni >= refill:
            take = min(ni, seq.size - cur)
            idx = np.flatnonzero(idle)[:take]
            rem[idx] = seq[cur:cur + take]
            cur += take
        act = rem > 0
        if not act.any():
            if cur >= seq.size:
                break
            continue
        # jump ahead until the next event (a lane finishing) to keep the model fast
        k = int(rem[act].min())
        if cur < seq.size:
            # a refill becomes possible when enough lanes are idle: step one finishing group at a time
            pass
        rem[act] -= k
        trips += k
        useful += k * int(act.sum())
    return trips, useful


def run(order_fn, nwaves=400, **kw):
    t = []
    for w in rng.choice(W, nwaves, replace=False):
        # wave w owns batches w, w+W, ... (64 rays each)
        ids = np.concatenate([np.arange((b * W + w) * 64, (b * W + w) * 64 + 64) for b in range(4)])
        ids = ids[ids < n]
        seq = order_fn(steps[ids])
        t.append(wave_trips(seq, **kw)[0])
    return np.mean(t)


ideal = steps.sum() / (W * 64)
print("ideal trips/wave %.1f" % ideal)
print("as generated, refill 16: %.1f" % run(lambda s: s))
print("as generated, refill 1:  %.1f" % run(lambda s: s, refill=1))
print("longest first (oracle knowledge), refill 16: %.1f" % run(lambda s: np.sort(s)[::-1]))
print("longest first, refill 1: %.1f" % run(lambda s: np.sort(s)[::-1], refill=1))
for noise in (0.5, 1.0, 2.0):
    def noisy(s, noise=noise):
        key = s + rng.normal(0, noise * 10.7, s.size)
        return s[np.argsort(-key)]
    print("longest first with predictor noise %.1f sigma: %.1f" % (noise, run(noisy)))
# 128 lanes' worth of rays per lane slot: what 8 rays per lane would give (4 waves per SIMD)
def run8(nwaves=200):
    t = []
    for w in rng.choice(W // 2, nwaves, replace=False):
        ids = np.concatenate([np.arange((b * (W // 2) + w) * 64, (b * (W // 2) + w) * 64 + 64) for b in range(8)])
        ids = ids[ids < n]
        t.append(wave_trips(steps[ids])[0])
    return np.mean(t)
print("8 rays per lane (half the waves): %.1f trips per wave = %.1f per 256 rays" % (run8(), run8() / 2))


# ---- workgroup model with ray donation in the drain (4 waves, donate to a lower-index sibling) ----
def wg_trips(seqs, refill=16, donate_max=16, donate=True, lanes=64):
    nw = len(seqs)
    rem = [np.zeros(lanes, dtype=np.int64) for _ in range(nw)]
    cur = [0] * nw
    alive = [True] * nw
    total = 0
    while any(alive):
        for w in range(nw):
            if not alive[w]:
                continue
            r = rem[w]
            idle = r == 0
            ni = int(idle.sum())
            if cur[w] < seqs[w].size and ni >= refill:
                take = min(ni, seqs[w].size - cur[w])
                idx = np.flatnonzero(idle)[:take]
                r[idx] = seqs[w][cur[w]:cur[w] + take]
                cur[w] += take
            elif donate and cur[w] >= seqs[w].size and w > 0:
                k = lanes - ni
                if 0 < k <= donate_max:
                    for j in range(w - 1, -1, -1):
                        if alive[j] and int((rem[j] == 0).sum()) >= k and cur[j] >= seqs[j].size:
                            idx = np.flatnonzero(rem[j] == 0)[:k]
                            rem[j][idx] = r[r > 0]
                            r[:] = 0
                            break
            if not (r > 0).any():
                if cur[w] >= seqs[w].size:
                    alive[w] = False
                continue
            r[r > 0] -= 1
            total += 1
    return total


def run_wg(nwg=150, **kw):
    t = []
    for g0 in rng.choice(W // 4, nwg, replace=False):
        seqs = []
        for w in range(4 * g0, 4 * g0 + 4):
            ids = np.concatenate([np.arange((b * W + w) * 64, (b * W + w) * 64 + 64) for b in range(4)])
            seqs.append(steps[ids[ids < n]])
        t.append(wg_trips(seqs, **kw) / 4.0)
    return np.mean(t)


print("WG model, no donation: %.1f trips per wave" % run_wg(donate=False))
for dm in (8, 16, 24, 32):
    print("WG model, donate at <= %d active: %.1f trips per wave" % (dm, run_wg(donate_max=dm)))
