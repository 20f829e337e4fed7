import itertools, random
groups = [[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],
          [4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
groups += [[l+32 for l in g] for g in groups]
def conflicts(f, stride_chunks, dxs=range(5)):
    tot = 0
    for dx in dxs:
        for grp in groups:
            seen = {}
            for l in grp:
                li, g = l % 16, l // 16
                hc = li + dx
                chunk = hc*stride_chunks + (g ^ f(hc))
                bq = chunk % 16
                seen[bq] = seen.get(bq,0)+1
            tot += sum(v-1 for v in seen.values())
    return tot
# candidates
print("none, stride4", conflicts(lambda h:0,4))
print("none, stride5", conflicts(lambda h:0,5))
print("h>>3&1<<1, stride4", conflicts(lambda h:((h>>3)&1)<<1,4))
print("(h>>2)&3, stride4", conflicts(lambda h:(h>>2)&3,4))
# brute force f over hc mod 16 -> 0..3 with stride 4 : 4^16 too many; random search / structured: f depends on (hc>>2)&3 only -> 4^4
best=None
for t in itertools.product(range(4),repeat=4):
    c=conflicts(lambda h:t[(h>>2)&3],4)
    if best is None or c<best[0]: best=(c,t)
print("best f((h>>2)&3)", best)
best=None
for t in itertools.product(range(4),repeat=8):
    c=conflicts(lambda h:t[(h>>1)&7],4)
    if best is None or c<best[0]: best=(c,t)
print("best f((h>>1)&7)", best)
