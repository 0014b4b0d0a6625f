"""Developer aid (GPU box): where oracle/rcp_model.h and v_rcp_f32 disagree, by input class."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as g
orc = g.load_oracle()
t = orc.refgpu_rcp_table()
bad, first = orc.refgpu_rcp_check(t, cap=4000000)
print("mismatches", bad, "captured", len(first))
x, hw, md = first[:, 0], first[:, 1], first[:, 2]
e = (x >> 23) & 0xFF
for ee in np.unique(e):
    m = e == ee
    print("input exponent field", int(ee), "count(captured)", int(m.sum()))
    idx = np.nonzero(m)[0]
    for i in list(idx[:6]) + list(idx[-6:]):
        print("   x=%08x hw=%08x model=%08x" % (x[i], hw[i], md[i]))
