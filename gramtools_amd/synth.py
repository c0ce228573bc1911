"""Synthetic PRGs and reads for the parity tests and bench.py (SURVEY.md §8d recipes).

No real genomes are available offline, so every workload is generated here with fixed seeds:
  * :func:`snp_prg`        random reference + SNP sites written as ``... 5 ref 6 alt 6 ...`` (the `normal`
                           mode of gramtools/commands/build/vcf_to_prg_string.py:81-101)
  * :func:`nested_prg`     small random bracket PRGs with nesting, empty alleles and adjacent sites
  * :func:`mixed_variant_prg` flat PRG with SNPs, indels, multi-allelic and adjacent sites
  * :func:`nested_regions_prg`  random sequence interleaved with nested bracket regions (the configs[2] recipe in small)
  * :func:`simulate_snp_reads` / :func:`simulate_graph_reads` / :func:`simulate_haplotype_reads`
                           error-free reads from random haplotypes
"""
import numpy as np


def random_ref(n: int, seed: int) -> np.ndarray:
    return np.random.default_rng(seed).integers(1, 5, size=n, dtype=np.uint8)


def snp_prg(ref: np.ndarray, n_sites: int, seed: int, min_gap: int = 2, multi_allelic_frac: float = 0.0):
    """Returns (prg_ints uint32, site_pos sorted ref positions, alts list-of-arrays).

    alt = ref base + 1 + (r mod 3) cyclic, i.e. always different from the reference base."""
    rng = np.random.default_rng(seed)
    G = ref.size
    # distinct positions at least min_gap apart: sample on a coarser grid then jitter
    cell = max(min_gap, G // max(n_sites, 1))
    if cell * n_sites > G:
        raise ValueError("too many sites for this reference")
    cells = np.sort(rng.choice(G // cell, size=n_sites, replace=False))
    pos = cells * cell + rng.integers(0, max(cell - min_gap + 1, 1), size=n_sites)
    pos = np.minimum(pos, G - 1)
    n_alts = np.ones(n_sites, dtype=np.int64)
    if multi_allelic_frac > 0:
        n_alts += (rng.random(n_sites) < multi_allelic_frac) * rng.integers(1, 3, size=n_sites)
    r = rng.integers(0, 3, size=n_sites)
    # per site: open marker, ref base, (separator, alt) x n_alts, closing marker
    per_site = 2 + 2 * n_alts
    out = np.empty(G + int(per_site.sum()), dtype=np.uint32)
    # positions of ref bases in the output
    shift = np.zeros(G, dtype=np.int64)
    np.add.at(shift, pos, per_site)
    ref_out = np.arange(G) + np.concatenate([[0], np.cumsum(shift)[:-1]])
    # ref base of a site sits after its opening marker
    site_open = ref_out[pos]
    ref_out_adj = ref_out.copy()
    ref_out_adj[pos] += 1
    out[ref_out_adj] = ref
    markers = 5 + 2 * np.arange(n_sites, dtype=np.uint32)
    out[site_open] = markers
    alts = []
    for a in range(int(n_alts.max())):
        sel = n_alts > a
        sep_pos = site_open[sel] + 2 + 2 * a
        out[sep_pos] = markers[sel] + 1
        alt = ((ref[pos[sel]].astype(np.int64) - 1 + 1 + (r[sel] + a) % 3) % 4 + 1).astype(np.uint32)
        out[sep_pos + 1] = alt
        alts.append((sel, alt))
    close_pos = site_open + 2 + 2 * n_alts
    out[close_pos] = markers + 1
    return out, pos, alts, n_alts


def simulate_snp_reads(ref, pos, alts, n_alts, n_reads: int, read_len: int, seed: int, alt_prob: float = 0.5,
                       rc_prob: float = 0.5):
    """Error-free reads from per-site Bernoulli haplotypes over a SNP PRG. Returns uint8 [n_reads, read_len]."""
    rng = np.random.default_rng(seed)
    G = ref.size
    starts = rng.integers(0, G - read_len + 1, size=n_reads)
    P = starts[:, None] + np.arange(read_len)[None, :]
    reads = ref[P]
    is_site = np.zeros(G, dtype=bool)
    is_site[pos] = True
    site_idx = np.zeros(G, dtype=np.int64)
    site_idx[pos] = np.arange(pos.size)
    take_alt = is_site[P] & (rng.random(P.shape) < alt_prob)
    if take_alt.any():
        rr, cc = np.nonzero(take_alt)
        s = site_idx[P[rr, cc]]
        which = (rng.random(s.size) * n_alts[s]).astype(np.int64)  # which alt allele
        alt_table = np.zeros((int(n_alts.max()), pos.size), dtype=np.uint8)
        for a, (sel, alt) in enumerate(alts):
            alt_table[a, sel] = alt
        reads[rr, cc] = alt_table[which, s]
    flip = rng.random(n_reads) < rc_prob
    if flip.any():
        reads[flip] = (5 - reads[flip])[:, ::-1]
    return np.ascontiguousarray(reads)


def flat_offsets(n_reads: int, read_len: int) -> np.ndarray:
    return (np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(read_len)).astype(np.uint64)


# ---------------------------------------------------------------------------------------------------
# small nested PRGs (tests)
# ---------------------------------------------------------------------------------------------------
def nested_prg(seed: int, n_top: int = 6, max_depth: int = 3, seq_max: int = 6, empty_allele_prob: float = 0.15,
               adjacent_prob: float = 0.2) -> str:
    """A random bracketed PRG string (``a[c,g[ct,t]a]c`` style)."""
    rng = np.random.default_rng(seed)

    def seq(lo, hi):
        n = int(rng.integers(lo, hi + 1))
        return "".join("acgt"[int(x)] for x in rng.integers(0, 4, size=n))

    def site(depth):
        n_alleles = int(rng.integers(2, 5))
        alleles = []
        for _ in range(n_alleles):
            if rng.random() < empty_allele_prob:
                alleles.append("")
                continue
            parts = [seq(0 if depth < max_depth and rng.random() < adjacent_prob else 1, seq_max)]
            if depth < max_depth and rng.random() < 0.4:
                parts.append(site(depth + 1))
                if rng.random() < adjacent_prob and depth + 1 <= max_depth:
                    parts.append(site(depth + 1))
                parts.append(seq(0 if rng.random() < adjacent_prob else 1, seq_max))
            alleles.append("".join(parts))
        if sum(1 for a in alleles if a == "") > 1:  # keep at most one empty allele per site
            first = True
            for i, a in enumerate(alleles):
                if a == "":
                    if first:
                        first = False
                    else:
                        alleles[i] = seq(1, seq_max)
        if all(a == "" for a in alleles):
            alleles[0] = seq(1, seq_max)
        return "[" + ",".join(alleles) + "]"

    out = [seq(2, seq_max * 2)]
    for _ in range(n_top):
        out.append(site(1))
        out.append(seq(0 if rng.random() < adjacent_prob else 1, seq_max * 2))
    out.append(seq(1, seq_max))
    return "".join(out)


def bracket_to_ints(s: str):
    """Bracketed PRG text -> ints, sites numbered by '[' order (linearised_prg.cpp:166-213 convention)."""
    base = {"a": 1, "c": 2, "g": 3, "t": 4, "A": 1, "C": 2, "G": 3, "T": 4}
    out, stack, mx = [], [], 3
    for ch in s:
        if ch == "[":
            mx += 2
            stack.append(mx)
            out.append(mx)
        elif ch == "]":
            out.append(stack.pop() + 1)
        elif ch == ",":
            out.append(stack[-1] + 1)
        else:
            out.append(base[ch])
    return np.asarray(out, dtype=np.uint32)


def simulate_graph_reads(prg_ints, n_reads: int, read_len: int, seed: int, rc_prob: float = 0.5, n_haps: int = 0):
    """Reads from random walks through an arbitrary (nested) PRG. Pure-Python: small cases only.

    A walk expands the PRG into one haplotype by picking a random allele at every site, then reads are
    substrings of haplotypes (a fresh haplotype per read, or one of `n_haps` pre-drawn haplotypes)."""
    rng = np.random.default_rng(seed)
    prg = [int(x) for x in prg_ints]
    n = len(prg)
    # matching structure: for each site-open position the list of allele (start, end) ranges and the close position
    open_info = {}
    stack = []
    for i, m in enumerate(prg):
        if m > 4 and m % 2 == 1:
            stack.append([i, [i + 1]])
        elif m > 4:
            top = stack[-1]
            # last occurrence of this even marker closes the site
            is_last = all(prg[j] != m for j in range(i + 1, n))
            if is_last:
                starts = top[1]
                ends = starts[1:] + [i + 1]
                open_info[top[0]] = ([(s, e - 1) for s, e in zip(starts, ends)], i)
                stack.pop()
            else:
                top[1].append(i + 1)

    def expand(lo, hi, out):
        i = lo
        while i < hi:
            m = prg[i]
            if m <= 4:
                out.append(m)
                i += 1
            else:
                alleles, close = open_info[i]
                s, e = alleles[int(rng.integers(0, len(alleles)))]
                expand(s, e, out)
                i = close + 1

    reads = []
    pool = []
    for _ in range(n_haps):
        hap = []
        expand(0, n, hap)
        pool.append(hap)
    for _ in range(n_reads):
        if pool:
            hap = pool[int(rng.integers(0, len(pool)))]
        else:
            hap = []
            expand(0, n, hap)
        L = min(read_len, len(hap))
        st = int(rng.integers(0, len(hap) - L + 1))
        r = np.asarray(hap[st:st + L], dtype=np.uint8)
        if rng.random() < rc_prob:
            r = (5 - r)[::-1]
        reads.append(np.ascontiguousarray(r))
    return reads


def mixed_variant_prg(ref: np.ndarray, n_sites: int, seed: int, max_alleles: int = 7, max_len: int = 6,
                      adjacent_prob: float = 0.05):
    """A flat (non-nested) PRG with SNPs, indels (alleles of 0..max_len bases, pure deletions included), multi-allelic
    sites (up to `max_alleles`, so both the dense grouped slots and the append log are used) and some adjacent sites.

    Returns (prg_ints uint32, sites) with sites = [(ref_start, ref_len, [allele arrays, allele 0 = reference])]."""
    rng = np.random.default_rng(seed)
    G = int(ref.size)
    sites = []
    p = int(rng.integers(1, 20))
    while len(sites) < n_sites and p + max_len + 2 < G:
        ref_len = int(rng.integers(1, max_len + 1)) if rng.random() < 0.3 else 1
        n_all = 2 if rng.random() < 0.8 else int(rng.integers(3, max_alleles + 1))
        alleles = [ref[p:p + ref_len].copy()]
        seen = {alleles[0].tobytes()}
        tries = 0
        while len(alleles) < n_all and tries < 50:
            tries += 1
            if ref_len == 1 and rng.random() < 0.7:
                a = rng.integers(1, 5, size=1, dtype=np.uint8)
            else:
                a = rng.integers(1, 5, size=int(rng.integers(0, max_len + 1)), dtype=np.uint8)
            if a.tobytes() in seen:
                continue
            seen.add(a.tobytes())
            alleles.append(a)
        sites.append((p, ref_len, alleles))
        gap = 0 if rng.random() < adjacent_prob else int(rng.integers(1, max(2, 2 * (G // max(n_sites, 1)))))
        p += ref_len + gap
    out = []
    cur = 0
    for i, (start, ref_len, alleles) in enumerate(sites):
        out.extend(int(x) for x in ref[cur:start])
        m = 5 + 2 * i
        out.append(m)
        for a in alleles:
            out.extend(int(x) for x in a)
            out.append(m + 1)
        cur = start + ref_len
    out.extend(int(x) for x in ref[cur:])
    return np.asarray(out, dtype=np.uint32), sites


def simulate_haplotype_reads(ref: np.ndarray, sites, n_reads: int, len_lo: int, len_hi: int, seed: int, n_haps: int = 16,
                             rc_prob: float = 0.5):
    """Error-free reads of ragged lengths in [len_lo, len_hi] from `n_haps` random haplotypes of a mixed_variant_prg."""
    rng = np.random.default_rng(seed)
    haps = []
    for _ in range(n_haps):
        parts, cur = [], 0
        for start, ref_len, alleles in sites:
            parts.append(ref[cur:start])
            parts.append(alleles[int(rng.integers(0, len(alleles)))])
            cur = start + ref_len
        parts.append(ref[cur:])
        haps.append(np.concatenate(parts).astype(np.uint8))
    reads = []
    for _ in range(n_reads):
        h = haps[int(rng.integers(0, n_haps))]
        L = int(min(rng.integers(len_lo, len_hi + 1), h.size))
        st = int(rng.integers(0, h.size - L + 1))
        r = h[st:st + L].copy()
        if rng.random() < rc_prob:
            r = (5 - r[::-1]).astype(np.uint8)
        reads.append(r)
    return reads


def nested_regions_prg(n_regions: int, seed: int, spacer_lo: int = 30, spacer_hi: int = 200):
    """Random sequence interleaved with `n_regions` nested bracket regions (depth <= 3, empty alleles, adjacent sites):
    BASELINE.json configs[2]'s "MSA regions" in small. Returns prg_ints (site markers renumbered 5, 7, 9, ...)."""
    rng = np.random.default_rng(seed)
    letters = "acgt"
    parts = []
    for r in range(n_regions):
        parts.append("".join(letters[int(x)] for x in rng.integers(0, 4, size=int(rng.integers(spacer_lo, spacer_hi)))))
        parts.append(nested_prg(seed * 1000 + r, n_top=int(rng.integers(1, 4)), max_depth=int(rng.integers(1, 4)),
                                seq_max=int(rng.integers(2, 8))))
    parts.append("".join(letters[int(x)] for x in rng.integers(0, 4, size=spacer_hi)))
    return bracket_to_ints("".join(parts))
