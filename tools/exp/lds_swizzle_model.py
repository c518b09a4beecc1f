"""LDS bank-conflict model of the packed large-frame kernel (stft_pk.h): 16 regions of MS complex, XOR-swizzled slots,
region stride RS.  Bank rules from MI355X_MICROARCH.md §LDS: ds_read_b64 = 2 groups of 32 lanes over 64 dword banks,
ds_write_b64 = 4 groups of 16 lanes over 32 dword banks.    python tools/exp/lds_swizzle_model.py [log2m]"""
import sys

from lds_conflicts import cycles


def sigma(e):
    return e ^ ((e >> 4) & 7) ^ (((e >> 6) & 1) << 3)


def model(log2m=13, rs_extra=2):
    M = 1 << log2m
    MS = M // 16
    TPFS = MS // 8
    RS = MS + rs_extra
    nw = MS // 64
    res, ideal = {}, {}

    def add(name, fn, write):
        for w in range(nw):
            a = [fn(w * 64 + l) for l in range(64)]
            res[name] = res.get(name, 0) + cycles(a, write)
            ideal[name] = ideal.get(name, 0) + (4 if write else 2)

    for k0 in range(16):
        add("W1 transpose", lambda t: k0 * RS + sigma(t), True)
    for r in range(2):
        reg = lambda t: (8 * r + t // TPFS) * RS
        for j in range(8):
            add("R gather", lambda t: reg(t) + sigma(t % TPFS + j * TPFS), False)
            add("W writeback", lambda t: reg(t) + sigma(t % TPFS + j * TPFS), True)
        p = 1
        npass = (log2m - 4 + 2) // 3
        for ps in range(npass - 1):
            for q in range(8):
                def wa(t, q=q, p=p):
                    si = t % TPFS
                    k = si & (p - 1)
                    return reg(t) + sigma((si - k) * 8 + k + q * p)
                add("W pass scatter", wa, True)
            for j in range(8):
                add("R pass gather", lambda t, j=j: reg(t) + sigma(t % TPFS + j * TPFS), False)
            p *= 8
    for q in range(8):
        def za(t, q=q, mirror=False):
            k = t + q * MS
            if mirror:
                k = (M - k) & (M - 1)
            return (k & 15) * RS + sigma(k >> 4)
        add("R unpack lo", za, False)
        add("R unpack hi", lambda t, q=q: za(t, q, True), False)
    return res, ideal, RS


if __name__ == "__main__":
    log2m = int(sys.argv[1]) if len(sys.argv) > 1 else 13
    for extra in (0, 1, 2, 4, 6, 10):
        res, ideal, RS = model(log2m, extra)
        print(f"RS = {RS}: total {sum(res.values())} vs {sum(ideal.values())} conflict-free;",
              {k: f"{res[k]}/{ideal[k]}" for k in res if res[k] != ideal[k]})
