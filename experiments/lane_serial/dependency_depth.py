import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from oracle_lib import Oracle
from k4os.compression.lz4_amd import corpus
o = Oracle()
def seqs(c):
    c = c.tolist(); n = len(c); p = 0; out = []; op = 0
    while p < n:
        t = c[p]; L = t >> 4; q = p + 1
        if L == 15:
            while True:
                e = c[q]; q += 1; L += e
                if e != 255: break
        q += L
        if q >= n: break
        off = c[q] | (c[q+1] << 8); q += 2
        ml = (t & 15) + 4
        if (t & 15) == 15:
            while True:
                e = c[q]; q += 1; ml += e
                if e != 255: break
        out.append((p, op, L, off, ml)); op += L + ml
        p = q
    return out
name = sys.argv[1]
d = corpus.class_bytes(name, 65536*2, 2)[65536:]
r, comp = o.compress_fast(d); c = np.array(comp[:r])
S = seqs(c)
# windows of 52*64 bytes starting at token positions (approx: fixed windows)
W = 52*64
wstart = 0; tot_steps = 0; nw = 0; idx = 0
while idx < len(S):
    wb = S[idx][0]
    lanes = [[] for _ in range(52)]
    j = idx
    while j < len(S) and S[j][0] < wb + W:
        lanes[(S[j][0]-wb)//64].append(S[j]); j += 1
    o0 = S[idx][1]
    base = []; 
    for l in lanes: base.append(l[0][1] if l else None)
    end = S[j][1] if j < len(S) else S[-1][1]+S[-1][2]+S[-1][4]
    # fill bases for empty lanes
    nb = end
    for k in range(51,-1,-1):
        if base[k] is None: base[k] = nb
        else: nb = base[k]
    base.append(end)
    prog = list(base[:52]); pos = [0]*52; have=[False]*52
    for k in range(52):
        if not lanes[k]: prog[k] = base[k+1]
    steps = 0
    while any(pos[k] < len(lanes[k]) for k in range(52)):
        steps += 1
        newprog = list(prog)
        for k in range(52):
            if pos[k] >= len(lanes[k]): continue
            p_, op_, L, off, ml = lanes[k][pos[k]]
            cur = op_ + L
            s = cur - off; e = min(s + ml, base[k])
            ready = True
            if e > o0 and s < base[k]:
                # owner of e-1
                own = max(i for i in range(52) if base[i] <= e-1)
                ready = prog[own] >= e
                s2 = max(s, o0)
                if ready and s2 < base[own]:
                    own0 = max(i for i in range(52) if base[i] <= s2)
                    for l2 in range(own0, own):
                        if prog[l2] < base[l2+1]: ready = False
            if ready:
                newprog[k] = cur + ml; pos[k] += 1
            else:
                newprog[k] = max(prog[k], cur)
        prog = newprog
    tot_steps += steps; nw += 1
    idx = j
print(name, "seqs", len(S), "windows", nw, "steps", tot_steps, "steps/window", tot_steps/nw, "max tokens/lane ~22")
# true data-dependency depth per window (byte-level): depth[byte] = 0 for literals / earlier windows; match byte = 1 + max depth of its source bytes (per sequence)
idx = 0; depths = []
while idx < len(S):
    wb = S[idx][0]; j = idx
    while j < len(S) and S[j][0] < wb + W: j += 1
    o0 = S[idx][1]
    end = S[j][1] if j < len(S) else S[-1][1]+S[-1][2]+S[-1][4]
    dep = np.zeros(end - o0 + 1, np.int32)
    mx = 0
    for (p_, op_, L, off, ml) in S[idx:j]:
        cur = op_ + L; s = cur - off
        d = 0
        lo = max(s, o0); hi = min(s + ml, cur)   # source bytes below own match start
        if hi > lo: d = int(dep[lo-o0:hi-o0].max())
        # overlapping part depends on itself: same depth
        dep[cur-o0:cur-o0+ml] = d + 1
        mx = max(mx, d + 1)
    depths.append(mx); idx = j
print(name, "true dependency depth per window:", depths)
