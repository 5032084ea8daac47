import sys, os, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from oracle_lib import Oracle
from k4os.compression.lz4_amd import corpus
o = Oracle()

def next_table(c):
    c = [int(v) for v in c]; n = len(c); nxt = [0]*n
    for p in range(n):
        t = c[p]; L = t >> 4; M = t & 15; q = p + 1
        if L == 15:
            while q < n:
                e = c[q]; q += 1; L += e
                if e != 255: break
        q2 = q + L
        x = q2 + 2
        ml = M + 4
        if M == 15:
            while x < n:
                e = c[x]; x += 1; ml += e
                if e != 255: break
        nxt[p] = x
    return nxt

def simulate(c, S=64, NL=52, EXTCAP=16, EXTRANGE=255):
    n = len(c); nxt = next_table(c)
    lim = n - 17   # chainable: p < lim
    ip = 0; rounds = 0; tot_main = 0; tot_ext = 0; tok = 0; hops = 0
    reasons = {}
    trueseq = 0
    p = 0
    while p < lim: trueseq += 1; p = nxt[p]
    while ip < lim:
        rounds += 1
        wb = ip; wend = wb + NL * S
        masks = [set() for _ in range(NL)]; x = [0]*NL; ext = [[] for _ in range(NL)]
        stop = [False]*NL
        maxmain = 0
        for j in range(NL):
            p = wb + j*S; se = p + S; k = 0
            if p >= lim: x[j] = p; stop[j] = True; continue
            while p < se:
                if p >= lim: stop[j] = True; break
                masks[j].add(p); p = nxt[p]; k += 1
            x[j] = p; maxmain = max(maxmain, k)
        maxext = 0
        status = [None]*NL
        for j in range(NL):
            p = x[j]; k = 0; ss = wb + j*S
            if stop[j]: status[j] = 'STOP'; continue
            while True:
                if p >= wend: status[j] = 'EXIT'; break
                if p >= lim: status[j] = 'STOP'; break
                jj = (p - wb)//S
                if p in masks[jj]: status[j] = 'MERGED'; break
                if k >= EXTCAP or p - ss > EXTRANGE: status[j] = 'UNMERGED'; break
                ext[j].append(p); p = nxt[p]; k += 1
            x[j] = p; maxext = max(maxext, k)
        # walk
        cur = 0; rin = wb; cnt = 0
        while True:
            hops += 1
            cnt += sum(1 for q in masks[cur] if q >= rin) + len(ext[cur])
            if status[cur] != 'MERGED': break
            rin = x[cur]; cur = (rin - wb)//S
        reasons[status[cur]] = reasons.get(status[cur], 0) + 1
        newip = x[cur]
        if cnt == 0: newip = nxt[ip]; cnt=0  # scalar
        tok += cnt; tot_main += maxmain; tot_ext += maxext
        ip = newip
    return dict(C=n, seq=trueseq, rounds=rounds, main=tot_main, ext=tot_ext, hops=hops, tok=tok, reasons=reasons)

if __name__ == '__main__':
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    NL = int(sys.argv[2]) if len(sys.argv) > 2 else 52
    CAP = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    for name in corpus.SILESIA_NAMES:
        d = corpus.class_bytes(name, 65536*2, 2)[65536:]
        r, comp = o.compress_fast(d)
        c = np.array(comp[:r])
        st = simulate(c, S, NL, CAP)
        steps = st['main'] + st['ext']
        est = steps*28 + st['hops']*4 + st['rounds']*400
        print(f"{name:8s} C={st['C']:6d} seq={st['seq']:5d} rounds={st['rounds']:3d} main={st['main']:4d} ext={st['ext']:4d} hops={st['hops']:4d} est_instr={est:6d} per_seq={est/max(1,st['seq']):.2f} {st['reasons']}")
