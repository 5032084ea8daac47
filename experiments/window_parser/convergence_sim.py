import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from oracle_lib import Oracle
from k4os.compression.lz4_amd import corpus
from conv import next_table
o = Oracle()

def simulate(c, S, NL, EXTCAP, EXTRANGE=255):
    n = len(c); nxt = next_table(c)
    lim = n - 17
    ip = 0; rounds = 0; tot_main = 0; tot_ext = 0; tok = 0; hops = 0; serial = 0; nserial_ev = 0
    trueseq = 0
    p = 0
    while p < lim: trueseq += 1; p = nxt[p]
    while ip < lim:
        rounds += 1
        wb = ip; wend = wb + NL * S
        marked = {}
        x = [0]*NL; stop = [False]*NL; nmain=[0]*NL
        maxmain = 0
        for j in range(NL):
            p = wb + j*S; se = p + S; k = 0
            if p >= lim: x[j] = p; stop[j] = True; continue
            while p < se:
                if p >= lim: stop[j] = True; break
                marked[p] = j; p = nxt[p]; k += 1
            x[j] = p; maxmain = max(maxmain, k)
        maxext = 0
        status = [None]*NL
        for j in range(NL):
            p = x[j]; k = 0; ss = wb + j*S
            if stop[j]: status[j] = 'STOP'; continue
            while True:
                if p >= wend: status[j] = 'EXIT'; break
                if p >= lim: status[j] = 'STOP'; break
                if p in marked: status[j] = 'MERGED'; break
                if k >= EXTCAP or p - ss > EXTRANGE: status[j] = 'UNMERGED'; break
                p = nxt[p]; k += 1
            x[j] = p; maxext = max(maxext, k)
        cur = 0
        while True:
            hops += 1
            st = status[cur]; xx = x[cur]
            if st == 'UNMERGED':
                nserial_ev += 1
                p = xx
                while True:
                    if p >= wend: st = 'EXIT'; break
                    if p >= lim: st = 'STOP'; break
                    if p in marked: st = 'MERGED'; break
                    p = nxt[p]; serial += 1
                xx = p
            if st != 'MERGED': break
            cur = marked[xx]
        newip = xx
        if newip == ip: newip = nxt[ip]
        tot_main += maxmain; tot_ext += maxext
        ip = newip
    return dict(C=n, seq=trueseq, rounds=rounds, main=tot_main, ext=tot_ext, hops=hops, serial=serial, nser=nserial_ev)

if __name__ == '__main__':
    S = int(sys.argv[1]); NL = int(sys.argv[2]); CAP = int(sys.argv[3])
    tot = 0; totseq = 0
    for name in corpus.SILESIA_NAMES:
        d = corpus.class_bytes(name, 65536*2, 2)[65536:]
        r, comp = o.compress_fast(d)
        c = np.array(comp[:r])
        st = simulate(c, S, NL, CAP)
        est = (st['main'] + st['ext'])*28 + st['hops']*4 + st['rounds']*400 + st['serial']*30
        tot += est; totseq += st['seq']
        print(f"{name:8s} C={st['C']:6d} seq={st['seq']:5d} rounds={st['rounds']:3d} main={st['main']:4d} ext={st['ext']:4d} hops={st['hops']:4d} serial={st['serial']:4d}/{st['nser']:3d} est={est:6d} per_seq={est/max(1,st['seq']):.2f}")
    print("total est", tot, "per seq", tot/totseq)
