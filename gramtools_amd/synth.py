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
import os

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


def simulate_snp_reads_fast(ref, pos, alts, n_alts, n_reads: int, read_len: int, seed: int, alt_prob: float = 0.5,
                            rc_prob: float = 0.5):
    """The read model of :func:`simulate_snp_reads` (uniform start, per-site Bernoulli alt, Bernoulli strand, error-free)
    with random draws only where a read meets a site: seconds per million reads instead of a quarter of a minute
    (bench.py cycles several distinct batches). Another random stream, so not the same reads for a given seed."""
    rng = np.random.default_rng(seed)
    G = ref.size
    starts = rng.integers(0, G - read_len + 1, size=n_reads)
    reads = np.lib.stride_tricks.sliding_window_view(ref, read_len)[starts]  # (n_reads, read_len) copy
    lo, hi = np.searchsorted(pos, starts), np.searchsorted(pos, starts + read_len)
    within, rr = _ragged_arange(hi - lo)
    s = lo[rr] + within
    take = rng.random(s.size) < alt_prob
    rr, s = rr[take], s[take]
    if s.size:
        which = (rng.random(s.size) * n_alts[s]).astype(np.int64)
        alt_table = np.zeros((int(n_alts.max()), pos.size), dtype=np.uint8)
        for a, (sel, alt) in enumerate(alts):
            alt_table[a, sel] = alt
        reads[rr, pos[s] - starts[rr]] = alt_table[which, s]
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


# ---------------------------------------------------------------------------------------------------
# vectorised recipes at BASELINE.json scale: configs[3] (SNPs + indels + multi-allelic sites, flat)
# and configs[2] (flat SNPs + nested "MSA regions"). Everything below is numpy index arithmetic: a 64 Mb / 1.8 M-site
# PRG and its haplotypes are built in seconds, so the same recipe serves an oracle-checked slice and the full size.
# ---------------------------------------------------------------------------------------------------
def _ragged_arange(lens):
    """[0..lens[0]), [0..lens[1]), ... concatenated, and the owner index of every element."""
    lens = np.asarray(lens, dtype=np.int64)
    total = int(lens.sum())
    owner = np.repeat(np.arange(lens.size, dtype=np.int64), lens)
    starts = np.cumsum(lens) - lens
    return np.arange(total, dtype=np.int64) - starts[owner], owner


class VariantSites:
    """Flat variant sites over a reference: site s replaces ref[start[s] : start[s] + ref_len[s]] by one of its alleles
    (allele 0 = the reference allele). Alleles are stored ragged: allele j of site s is
    bases[a_off[a_first[s] + j] : a_off[a_first[s] + j + 1]]."""

    def __init__(self, start, ref_len, n_alleles, a_first, a_off, bases):
        self.start, self.ref_len, self.n_alleles = start, ref_len, n_alleles
        self.a_first, self.a_off, self.bases = a_first, a_off, bases

    @property
    def n_sites(self):
        return int(self.start.size)

    def allele(self, s, j):
        a = int(self.a_first[s]) + j
        return self.bases[int(self.a_off[a]):int(self.a_off[a + 1])]


def variant_sites(ref: np.ndarray, n_sites: int, seed: int, snp_frac: float = 0.9, multi_frac: float = 0.05,
                  max_indel: int = 10) -> VariantSites:
    """The configs[3] site mix: `snp_frac` SNPs, the rest 1..max_indel bp indels in three equal kinds (insertion behind an
    anchor base, deletion behind an anchor base, pure deletion = an empty alternative allele); independently
    `multi_frac` of all sites carry 3-4 alleles. Alleles of one site are pairwise different. One site per grid cell of
    G // n_sites bases, so sites never touch (cells must be >= max_indel + 3 bases when indels are requested)."""
    rng = np.random.default_rng(seed)
    G = int(ref.size)
    cell = G // max(n_sites, 1)
    span = max_indel + 1 if snp_frac < 1.0 else 1
    if cell < span + 2:
        raise ValueError("too many sites for this reference")
    start = np.arange(n_sites, dtype=np.int64) * cell + rng.integers(1, cell - span, size=n_sites)
    kind = np.where(rng.random(n_sites) < snp_frac, 0, rng.integers(1, 4, size=n_sites))  # 0 SNP, 1 ins, 2 del, 3 pure del
    d = rng.integers(1, max_indel + 1, size=n_sites)
    ref_len = np.where(kind == 0, 1, np.where(kind == 1, 1, np.where(kind == 2, 1 + d, d))).astype(np.int64)
    n_alleles = np.where(rng.random(n_sites) < multi_frac, rng.integers(3, 5, size=n_sites), 2).astype(np.int64)
    a_first = np.cumsum(n_alleles) - n_alleles
    j, site = _ragged_arange(n_alleles)                       # allele index within its site, owning site
    k, dd = kind[site], d[site]
    # lengths: allele 0 = ref_len; SNP alts 1; insertion alts 1 + d (+ j - 1 for the extra ones: pairwise different
    # lengths); deletion-with-anchor alts 1, 2 + d, 3 + d; pure deletion alts 0, d + 1, d + 2
    alt_len = np.where(k == 0, 1,
               np.where(k == 1, dd + j,
               np.where(k == 2, np.where(j == 1, 1, dd + j),
                        np.where(j == 1, 0, dd + j - 1))))
    a_len = np.where(j == 0, ref_len[site], alt_len).astype(np.int64)
    a_off = np.concatenate([[0], np.cumsum(a_len)]).astype(np.int64)
    within, owner = _ragged_arange(a_len)                      # base index within its allele, owning allele
    bases = rng.integers(1, 5, size=int(a_len.sum()), dtype=np.uint8)
    o_site, o_j, o_kind = site[owner], j[owner], k[owner]
    is_ref = o_j == 0
    bases[is_ref] = ref[start[o_site[is_ref]] + within[is_ref]]
    snp_alt = (o_kind == 0) & ~is_ref                          # the j-th alternative of a SNP: ref base + j (cyclic)
    bases[snp_alt] = ((ref[start[o_site[snp_alt]]].astype(np.int64) - 1 + o_j[snp_alt]) % 4 + 1).astype(np.uint8)
    anchored = ((o_kind == 1) | (o_kind == 2)) & ~is_ref & (within == 0)  # VCF-style indels keep the anchor base
    bases[anchored] = ref[start[o_site[anchored]]]
    # a deletion's alternative must differ from the reference allele even when equally long: impossible by the lengths above
    return VariantSites(start, ref_len, n_alleles, a_first, a_off, bases)


def _splice(ref: np.ndarray, start, ref_len, piece_off, piece_bases, dtype):
    """ref with [start[s], start[s] + ref_len[s]) replaced by piece s = piece_bases[piece_off[s] : piece_off[s + 1]].
    Returns (out, pos_of_ref) where pos_of_ref[i] = position in `out` of reference base i (of the first base at or
    after i that is kept, for bases inside a replaced span)."""
    G = int(ref.size)
    piece_len = np.diff(piece_off)
    keep = np.ones(G, dtype=bool)
    w, owner = _ragged_arange(ref_len)
    keep[start[owner] + w] = False
    grow = np.zeros(G + 1, dtype=np.int64)                     # extra output symbols emitted before reference base i
    np.add.at(grow, start, piece_len)
    pos_of_ref = np.cumsum(keep) - keep + np.cumsum(grow[:G])  # kept bases before i + pieces starting at or before i
    out = np.empty(int(keep.sum() + piece_len.sum()), dtype=dtype)
    out[pos_of_ref[keep]] = ref[keep]
    pw, powner = _ragged_arange(piece_len)
    piece_start = pos_of_ref[start] - piece_len                # a piece sits right before the first kept base after it
    out[piece_start[powner] + pw] = piece_bases[piece_off[powner] + pw]
    return out, pos_of_ref


def variant_prg(ref: np.ndarray, sites: VariantSites):
    """The PRG of `sites` over `ref`: ... 5 allele0 6 allele1 6 [allele2 6 ...] ... (vcf_to_prg_string.py:81-101).
    Returns (prg uint32, pos_of_ref)."""
    S = sites.n_sites
    a_len = np.diff(sites.a_off)
    n_all = sites.n_alleles
    blk_len = 1 + np.add.reduceat(a_len, sites.a_first) + n_all if S else np.zeros(0, dtype=np.int64)
    blk_off = np.concatenate([[0], np.cumsum(blk_len)]).astype(np.int64)
    blk = np.empty(int(blk_off[-1]), dtype=np.uint32)
    marker = (5 + 2 * np.arange(S, dtype=np.int64)).astype(np.uint32)
    blk[blk_off[:-1]] = marker
    # allele a of site s starts at blk_off[s] + 1 + (bases of earlier alleles of s) + (its index in the site)
    j, site = _ragged_arange(n_all)
    before = (sites.a_off[:-1] - sites.a_off[sites.a_first[site]]) + j
    a_start = blk_off[site] + 1 + before
    w, owner = _ragged_arange(a_len)
    blk[a_start[owner] + w] = sites.bases[sites.a_off[owner] + w]
    blk[a_start + a_len] = marker[site] + 1                    # separator behind every allele; the last one closes the site
    return _splice(ref.astype(np.uint32), sites.start, sites.ref_len, blk_off, blk, np.uint32)


def variant_haplotype(ref: np.ndarray, sites: VariantSites, seed: int, alt_prob: float = 0.5):
    """One haplotype: at every site the reference allele with probability 1 - alt_prob, else a uniform alternative.
    Returns (bases uint8, pos_of_ref)."""
    rng = np.random.default_rng(seed)
    S = sites.n_sites
    pick = np.where(rng.random(S) < alt_prob, 1 + (rng.random(S) * (sites.n_alleles - 1)).astype(np.int64), 0)
    a = sites.a_first + pick
    a_len = sites.a_off[a + 1] - sites.a_off[a]
    off = np.concatenate([[0], np.cumsum(a_len)]).astype(np.int64)
    w, owner = _ragged_arange(a_len)
    pieces = sites.bases[sites.a_off[a][owner] + w]
    return _splice(ref, sites.start, sites.ref_len, off, pieces, np.uint8)


def reads_from_haplotypes(haps, n_reads: int, read_len: int, seed: int, rc_prob: float = 0.5) -> np.ndarray:
    """Error-free reads: a uniform haplotype, a uniform start, either strand. uint8 [n_reads, read_len]."""
    rng = np.random.default_rng(seed)
    which = rng.integers(0, len(haps), size=n_reads)
    reads = np.empty((n_reads, read_len), dtype=np.uint8)
    cols = np.arange(read_len)[None, :]
    for h, hap in enumerate(haps):
        sel = np.nonzero(which == h)[0]
        st = rng.integers(0, hap.size - read_len + 1, size=sel.size)
        reads[sel] = hap[st[:, None] + cols]
    flip = rng.random(n_reads) < rc_prob
    reads[flip] = (5 - reads[flip])[:, ::-1]
    return reads


def realistic_reads(reads2d: np.ndarray, seed: int, sub_rate: float = 0.005, n_read_frac: float = 0.01,
                    len_lo: int = 100, len_hi: int = 0):
    """Error-free simulated reads -> reads as a sequencer delivers them: every base substituted with
    probability `sub_rate` (by one of the three other bases), `n_read_frac` of the reads with one to three `N`s (byte 0:
    the whole read is skipped, common/utils.cpp:73-92), and ragged lengths — every read cut at its 3' end to a length drawn
    from [len_lo, len_hi] (len_hi = 0: the input length). Returns (flat uint8, offsets uint64). Vectorised."""
    rng = np.random.default_rng(seed)
    r = np.array(reads2d, dtype=np.uint8, copy=True)
    n, L = r.shape
    hi = len_hi or L
    assert 1 <= len_lo <= hi <= L
    sub = rng.random(r.shape) < sub_rate
    shift = rng.integers(1, 4, size=int(sub.sum()), dtype=np.uint8)
    r[sub] = (r[sub] - 1 + shift) % 4 + 1
    lens = rng.integers(len_lo, hi + 1, size=n).astype(np.int64)
    with_n = np.flatnonzero(rng.random(n) < n_read_frac)
    for _ in range(3):  # one to three Ns inside the kept part of the read
        take = with_n[rng.random(with_n.size) < (1.0 if _ == 0 else 0.5)]
        r[take, (rng.random(take.size) * lens[take]).astype(np.int64)] = 0
    keep = np.arange(L)[None, :] < lens[:, None]
    offs = np.zeros(n + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens)
    return np.ascontiguousarray(r[keep]), offs


def split_reads(flat: np.ndarray, offs: np.ndarray):
    """(flat, offsets) -> list of per-read arrays (views)."""
    o = offs.astype(np.int64)
    return [flat[o[i]:o[i + 1]] for i in range(o.size - 1)]


def chr20_recipe(G: int, n_sites: int, n_reads: int, seed: int, n_haps: int = 4, read_len: int = 150):
    """BASELINE.json configs[3] at any size: random reference of G bases, n_sites sites (90 % SNPs, 10 % 1-10 bp indels
    incl. pure deletions, 5 % with 3-4 alleles; full size: G = 64 444 167, 1.8 M sites, k = 14), error-free reads
    from `n_haps` haplotypes. Returns (prg, reads)."""
    ref = random_ref(G, seed)
    sites = variant_sites(ref, n_sites, seed + 1)
    prg, _ = variant_prg(ref, sites)
    haps = [variant_haplotype(ref, sites, seed + 10 + h)[0] for h in range(n_haps)]
    return prg, reads_from_haplotypes(haps, n_reads, read_len, seed + 2)


def _genome_chunk(job):
    """One chunk of genome_recipe_file (a worker process): its piece of the PRG with the site markers renumbered, and its
    share of the reads."""
    g, ns, nr, site0, seed, n_haps, read_len = job
    ref = random_ref(g, seed)
    sites = variant_sites(ref, ns, seed + 1)
    prg, _ = variant_prg(ref, sites)
    marker = prg > 4
    prg[marker] += np.uint32(2 * site0)
    haps = [variant_haplotype(ref, sites, seed + 10 + h)[0] for h in range(n_haps)]
    reads = reads_from_haplotypes(haps, nr, read_len, seed + 2) if nr else np.zeros((0, read_len), dtype=np.uint8)
    return prg, reads


def genome_recipe_file(path: str, G: int, n_sites: int, n_reads: int, seed: int, chunk: int = 50_000_000, n_haps: int = 2,
                       read_len: int = 150, workers: int = 0):
    """BASELINE.json configs[4] (3.1 G bases, 85 M sites "same mix" as configs[3]): chr20_recipe's site mix —
    90 % SNPs, 10 % 1-10 bp indels incl. pure deletions, 5 % of the sites with 3-4 alleles — over G bases, generated chunk by
    chunk in worker processes and written to `path` as gram_dir/prg (little-endian uint32). Reads are error-free draws from
    `n_haps` haplotypes per chunk (none crosses a chunk boundary), shuffled. Returns (n_symbols, reads uint8 [n_reads, L])."""
    import multiprocessing as mp
    n_chunks = (G + chunk - 1) // chunk
    jobs, site0, read0 = [], 0, 0
    for c in range(n_chunks):
        g = min(chunk, G - c * chunk)
        ns = int(round(n_sites * (c * chunk + g) / G)) - site0
        nr = int(round(n_reads * (c * chunk + g) / G)) - read0
        jobs.append((g, ns, nr, site0, seed + 100 * c, n_haps, read_len))
        site0 += ns
        read0 += nr
    workers = workers or max(1, min(n_chunks, (os.cpu_count() or 8), 16))
    n_symbols, reads = 0, []
    with open(path, "wb") as f:
        if workers == 1 or n_chunks == 1:
            results = map(_genome_chunk, jobs)
            for prg, r in results:
                prg.astype("<u4", copy=False).tofile(f)
                n_symbols += int(prg.size)
                reads.append(r)
        else:
            with mp.get_context("fork").Pool(workers) as pool:
                for prg, r in pool.imap(_genome_chunk, jobs):
                    prg.astype("<u4", copy=False).tofile(f)
                    n_symbols += int(prg.size)
                    reads.append(r)
    reads = np.concatenate(reads) if reads else np.zeros((0, read_len), dtype=np.uint8)
    np.random.default_rng(seed + 7).shuffle(reads, axis=0)
    return n_symbols, reads


def msa_region(rng, target_len: int, max_depth: int = 3) -> str:
    """One "MSA region" of the configs[2] recipe as a bracketed PRG: about `target_len` bases along its first alleles,
    sites with 2-6 alleles, nesting up to `max_depth`, empty alleles and adjacent sites."""
    def seq(lo, hi):
        return "".join("acgt"[int(x)] for x in rng.integers(0, 4, size=int(rng.integers(lo, hi + 1))))

    def site(depth):
        n_alleles = int(rng.integers(2, 7))
        alleles, seen = [], set()
        while len(alleles) < n_alleles:
            if rng.random() < 0.1 and "" not in seen:
                a = ""
            else:
                a = seq(1, 12)
                if depth < max_depth and rng.random() < 0.3:
                    a += site(depth + 1)
                    if rng.random() < 0.2:
                        a += site(depth + 1)           # adjacent sites inside an allele
                    a += seq(0 if rng.random() < 0.2 else 1, 8)
            if a in seen:
                continue
            seen.add(a)
            alleles.append(a)
        return "[" + ",".join(alleles) + "]"

    out, n = [], 0
    while n < target_len:
        s = seq(3, 25)
        out.append(s)
        out.append(site(1))
        n += len(s) + 6
        if rng.random() < 0.15:
            out.append(site(1))                        # adjacent top-level sites
            n += 6
    return "".join(out)


def _expand_region(ints, rng):
    """One random haplotype through a (nested) region given as PRG ints."""
    prg = [int(x) for x in ints]
    n = len(prg)
    last = {}
    for i, m in enumerate(prg):
        if m > 4 and m % 2 == 0:
            last[m] = i
    open_info, stack = {}, []
    for i, m in enumerate(prg):
        if m > 4 and m % 2 == 1:
            stack.append([i, [i + 1]])
        elif m > 4:
            top = stack[-1]
            if last[m] == i:
                starts = top[1]
                ends = starts[1:] + [i + 1]
                open_info[top[0]] = ([(s, e - 1) for s, e in zip(starts, ends)], i)
                stack.pop()
            else:
                top[1].append(i + 1)
    out = []

    def expand(lo, hi):
        i = lo
        while i < hi:
            m = prg[i]
            if m <= 4:
                out.append(m)
                i += 1
            else:
                alleles, close = open_info[i]
                s, e = alleles[int(rng.integers(0, len(alleles)))]
                expand(s, e)
                i = close + 1

    expand(0, n)
    return np.asarray(out, dtype=np.uint8)


def renumber_markers(prg: np.ndarray) -> np.ndarray:
    """Site markers renumbered 5, 7, 9, ... in order of appearance of the opening (odd) marker."""
    prg = np.asarray(prg, dtype=np.uint32)
    is_open = (prg > 4) & (prg % 2 == 1)
    opens = prg[is_open]
    table = np.zeros(int(prg.max()) + 2, dtype=np.uint32)
    table[opens] = 5 + 2 * np.arange(opens.size, dtype=np.uint32)
    table[opens + 1] = table[opens] + 1
    out = prg.copy()
    mk = prg > 4
    out[mk] = table[prg[mk]]
    return out


def pf3d7_recipe(G: int, n_regions: int, n_snps: int, n_reads: int, seed: int, n_haps: int = 4, read_len: int = 150,
                 region_lo: int = 200, region_hi: int = 800):
    """BASELINE.json configs[2] at any size: random reference of G bases with n_snps flat SNP sites, and n_regions nested
    "MSA regions" of region_lo..region_hi bases (depth <= 3, 2-6 alleles per site, empty alleles, adjacent sites)
    inserted between reference bases (full size: G = 23.3 Mb, 2 000 regions, 100 k SNPs, k = 10). Error-free reads from
    `n_haps` haplotypes. Returns (prg, reads)."""
    rng = np.random.default_rng(seed)
    ref = random_ref(G, seed + 1)
    sites = variant_sites(ref, n_snps, seed + 2, snp_frac=1.0, multi_frac=0.05)
    backbone, pos_of_ref = variant_prg(ref, sites)
    # insertion points: reference positions that are not a SNP, at least 300 bases apart
    cell = G // max(n_regions, 1)
    at = np.arange(n_regions, dtype=np.int64) * cell + rng.integers(10, max(cell - 300, 11), size=n_regions)
    is_site = np.zeros(G, dtype=bool)
    is_site[sites.start] = True
    at += is_site[at]                                          # SNPs never touch: the next base is plain
    regions = [bracket_to_ints(msa_region(rng, int(rng.integers(region_lo, region_hi + 1)))) for _ in range(n_regions)]
    # shift the regions' markers past the backbone's, then renumber everything by appearance
    base_marker = 5 + 2 * sites.n_sites
    parts, cur = [], 0
    for r, ints in enumerate(regions):
        cut = int(pos_of_ref[at[r]])
        parts.append(backbone[cur:cut])
        shifted = ints.copy()
        mk = shifted > 4
        shifted[mk] += np.uint32(base_marker - 5)
        base_marker += 2 * int(((ints > 4) & (ints % 2 == 1)).sum())
        parts.append(shifted)
        cur = cut
    parts.append(backbone[cur:])
    prg = renumber_markers(np.concatenate(parts))
    haps = []
    for h in range(n_haps):
        hap, hpos = variant_haplotype(ref, sites, seed + 10 + h)
        hrng = np.random.default_rng(seed + 100 + h)
        hp, cur = [], 0
        for r, ints in enumerate(regions):
            cut = int(hpos[at[r]])
            hp.append(hap[cur:cut])
            hp.append(_expand_region(ints, hrng))
            cur = cut
        hp.append(hap[cur:])
        haps.append(np.concatenate(hp))
    return prg, reads_from_haplotypes(haps, n_reads, read_len, seed + 3)
