import itertools
G128 = [list(range(0,4))+list(range(12,16))+list(range(20,28)), list(range(4,12))+list(range(16,20))+list(range(28,32))]
def read_conf(f, shift):
    # extra cycles over all 4 groups (two per half-wave, pieces differ between halves but only relative matters within group: same piece)
    tot = 0
    for grp in G128:
        banks = {}
        for l in grp:
            row = l + shift
            key = (row % 4, f(row))
            banks[key] = banks.get(key, 0) + 1
        tot += max(banks.values()) - 1
    return tot
def write_conf(f, start):
    banks = {}
    for i in range(8):
        row = start + i
        key = (row % 2, f(row))
        banks[key] = banks.get(key, 0) + 1
    return max(banks.values()) - 1
NB = 7
best = []
for m0 in range(1 << NB):
    for m1 in range(1 << NB):
        def f(row, m0=m0, m1=m1):
            return (bin(row & m0).count("1") & 1) | ((bin(row & m1).count("1") & 1) << 1)
        r0 = sum(read_conf(f, s) for s in (0, 32, 64, 96))
        r1 = sum(read_conf(f, s) for s in (1, 33, 65, 97))
        w1 = sum(write_conf(f, s) for s in range(0, 128, 8))
        w2 = sum(write_conf(f, s) for s in range(0, 128))
        best.append((r0 + r1, w1, w2, r0, r1, m0, m1))
best.sort()
for b in best[:10]: print(b)
cur = lambda row: (row >> 2) & 3
print("current", sum(read_conf(cur, s) for s in (0,32,64,96)), sum(read_conf(cur, s) for s in (1,33,65,97)), sum(write_conf(cur, s) for s in range(0,128,8)), sum(write_conf(cur, s) for s in range(0,128)))
# best with zero read conflicts, minimal write
z = [b for b in best if b[0] == 0]
z.sort(key=lambda b: (b[1] + b[2]))
print(z[:10])
