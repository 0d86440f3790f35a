"""Host-side model of the stream-K work split of the grouped filter-gradient launch (csrc/conv_wgrad.hip: conv_wgrad_*_sk_kernel
and wgrad_sk_fixup_kernel use exactly this integer arithmetic).  Checked on random groups: every (problem, tile) K range is covered
once, without gaps, by consecutive workgroups; a workgroup leaves at most one 'continues a tile' partial (slot 0) and one 'begins a
tile' partial (slot 1); and the plan entry of a tile's last contributor (first workgroup, slot sequence) names exactly the partials that
were written, in workgroup order (the fixed summation order that makes the result reproducible)."""
import random


def _split(units, max_g, min_units=2):
    prefix, tot = [], 0
    for tiles, T in units:
        tot += tiles * T
        prefix.append(tot)
    G = max(1, min(max_g, tot // min_units))
    written = {}
    for w in range(G):
        u, end, idx, slots_used = w * tot // G, (w + 1) * tot // G, 0, []
        while u < end:
            while u >= prefix[idx]:
                idx += 1
            base = prefix[idx - 1] if idx else 0
            T = units[idx][1]
            tile = (u - base) // T
            k0 = (u - base) - tile * T
            k1 = min(T, k0 + (end - u))
            full = k0 == 0 and k1 == T
            slot = None if full else (0 if k0 > 0 else 1)
            if slot is not None:
                assert slot not in slots_used, "a workgroup may use each slot once"
                slots_used.append(slot)
            written.setdefault((idx, tile), []).append((slot, w, k0, k1))
            u += k1 - k0
    # plan entries (sk_write_plan): workgroup w is a tile's LAST contributor when its first segment continues the tile (k0 > 0)
    # and ends it (k1 == T); the tile's first workgroup wf = the largest w' with w' * U / G <= tile start
    fixed = {}
    for w in range(G):
        u, end, idx = w * tot // G, (w + 1) * tot // G, 0
        while u >= prefix[idx]:
            idx += 1
        base = prefix[idx - 1] if idx else 0
        T = units[idx][1]
        tile = (u - base) // T
        k0 = (u - base) - tile * T
        k1 = min(T, k0 + (end - u))
        if not (k0 > 0 and k1 == T):
            continue
        ts = u - k0
        wf = ts * G // tot
        while (wf + 1) * tot // G <= ts:
            wf += 1
        while wf * tot // G > ts:
            wf -= 1
        assert (idx, tile) not in fixed, "two last contributors for one tile"
        fixed[(idx, tile)] = [(1, wf)] + [(0, v) for v in range(wf + 1, w + 1)]
    return written, fixed


def test_streamk_ranges_cover_and_fixup_matches():
    rng = random.Random(1)
    for _ in range(1500):
        units = [(rng.randint(1, 30), rng.choice([1, 2, 3, 4, 7, 16, 64, 128, 333, 4096])) for _ in range(rng.randint(1, 12))]
        written, fixed = _split(units, rng.choice([1, 2, 3, 7, 64, 512, 768, 1280]), rng.choice([2, 4, 16]))
        for idx, (tiles, T) in enumerate(units):
            for t in range(tiles):
                parts = written[(idx, t)]
                ranges = sorted((k0, k1) for _, _, k0, k1 in parts)
                assert ranges[0][0] == 0 and ranges[-1][1] == T and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
                if parts[0][0] is None:
                    assert len(parts) == 1 and (idx, t) not in fixed
                else:
                    assert [(s, w) for s, w, _, _ in parts] == fixed[(idx, t)]
