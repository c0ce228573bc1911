"""The `gram` drop-in executable: argv/exit-code contract on CPU, end-to-end IT1-IT3 runs on the GPU."""
import gzip
import json
import os
import subprocess

import numpy as np
import pytest

from golden_runner import load_cases
from gramtools_amd.build import build_gram
from oracle import ints_to_prg_bytes

GRAM = build_gram()


def run(*args):
    # the Python front-end passes only LD_LIBRARY_PATH in the child environment (gramtools/commands/common.py:33-49)
    return subprocess.run([GRAM, *args], capture_output=True, text=True, env={"LD_LIBRARY_PATH": ""})


def test_no_arguments_prints_help_and_exits_zero():
    r = run()
    assert r.returncode == 0 and "command to execute" in r.stdout  # probed by gramtools_main.py:73-90


def test_unknown_command_exits_one():
    r = run("frobnicate")
    assert r.returncode == 1 and "Unrecognised command" in r.stdout


def test_build_validates_prg(tmp_path):
    (tmp_path / "prg").write_bytes(ints_to_prg_bytes([1, 5, 2, 6, 3, 6, 4]))
    assert run("build", "--gram_dir", str(tmp_path), "--kmer_size", "2").returncode == 0
    (tmp_path / "prg").write_bytes(ints_to_prg_bytes([5, 1, 6, 2, 6, 2, 5, 1, 6, 3, 6]))  # duplicate site marker
    assert run("build", "--gram_dir", str(tmp_path), "--kmer_size", "2").returncode == 1


def _it_cases():
    return [c for c in load_cases("graph_and_kmers.json") if c["name"].startswith("IT")]


def _write_fastq(path, reads, gz=False):
    txt = "".join(f"@r{i}\n{r}\n+\n{'5' * len(r)}\n" for i, r in enumerate(reads))
    if gz:
        with gzip.open(path, "wt") as fh:
            fh.write(txt)
    else:
        path.write_text(txt)


@pytest.mark.gpu
@pytest.mark.parametrize("case", _it_cases(), ids=[c["name"] for c in _it_cases()])
@pytest.mark.parametrize("gz", [False, True])
def test_integration_cases_through_the_cli(tmp_path, case, gz):
    """gramtools/tests/genotype/test_genotype_integration_tests.py:68-157 through `gram genotype`."""
    (tmp_path / "prg").write_bytes(ints_to_prg_bytes(case["prg"]["ints"]))
    reads = [op for op in case["ops"] if op["op"] == "map_reads"][0]["reads"]
    fq = tmp_path / ("reads.fastq.gz" if gz else "reads.fastq")
    _write_fastq(fq, reads, gz)
    out = tmp_path / "run"
    r = run("genotype", "--gram_dir", str(tmp_path), "--reads", str(fq), "--sample_id", "test", "--ploidy", "haploid",
            "--kmer_size", "5", "--genotype_dir", str(out), "--max_threads", "1", "--seed", "42")
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Count all reads: 4" in r.stdout and "Count exact mapped reads: 2" in r.stdout
    pb = json.loads((out / "coverage" / "allele_base_coverage.json").read_text())["allele_base_counts"]
    gp = json.loads((out / "coverage" / "grouped_allele_counts_coverage.json").read_text())["grouped_allele_counts"]
    for op in case["ops"]:
        if op["op"] == "expect_allele_base":
            assert pb == op["value"]
        if op["op"] == "expect_grouped":
            groups = {gid: ",".join(str(a) for a in ids) for gid, ids in gp["allele_groups"].items()}
            got = [{groups[g]: c for g, c in site.items()} for site in gp["site_counts"]]
            assert got == op["value"]
    asum = [[int(x) for x in line.split()] for line in (out / "coverage" / "allele_sum_coverage").read_text().splitlines()]
    assert len(asum) == len(gp["site_counts"])
    rs = json.loads((out / "read_stats.json").read_text())
    assert rs["Max_read_length"] == max(len(x) for x in reads)
    assert abs(rs["Quality"]["Error_rate_mean"] - 0.01) < 1e-9 and rs["Quality"]["Num_bases"] == sum(len(x) for x in reads)
    # infer stage (genotype.cpp:72-118): the three files the unmodified front-end reads afterwards (genotype.py:131-145)
    geno = out / "genotype"
    j = json.loads((geno / "genotyped.json").read_text())
    assert j["Model"] == "LevelGenotyping" and len(j["Sites"]) == len(gp["site_counts"])
    vcf = gzip.decompress((geno / "genotyped.vcf.gz").read_bytes()).decode().splitlines()
    assert vcf[0] == "##fileformat=VCFv4.2" and vcf[-1].split("\t")[0] == "gramtools_prg"
    fa = (geno / "personalised_reference.fasta").read_text().splitlines()
    assert fa[0].startswith(">gramtools_prg test personalised reference made by gramtools genotype") and set("".join(fa[1:])) <= set("ACGT")
    # the CALLS: every site of genotyped.json against the pinned infer routines (tests/test_infer*.py) run on the coverage
    # of the host emulation of the device logic (== the oracle's, tests/test_hostemu_parity.py) with the same statistics
    import numpy as np
    from common import hostemu_map
    from golden_runner import seq
    from gramtools_amd import Index, Coverage, QuasimapReadsStats, Genotyped, master_seeds
    prg = np.asarray(case["prg"]["ints"], dtype=np.uint32)
    rd = [seq(r) for r in reads]
    raw, _, rc = hostemu_map(prg, 5, rd, master_seeds(42, [len(rd)]), return_raw=True)
    assert rc == 0
    ix = Index(prg, 5, threads=1)
    cov = Coverage(ix, raw["allele_sum"], raw["per_base"], raw["grouped"], raw["grouped_log"], QuasimapReadsStats(*(int(x) for x in raw["stats"])))
    # (read_stats.json holds 6 significant digits, as ReadStats::serialise's stream does, read_stats.cpp:162-209; the
    # executable genotypes with the doubles themselves: the depth of this coverage, 10^-(mean phred)/10 of the '5's)
    import math
    d = cov.depth_stats()
    assert abs(rs["Read_depth"]["Mean"] - d["mean"]) <= 1e-5 * max(1, d["mean"])
    assert abs(rs["Read_depth"]["Variance"] - d["variance"]) <= 1e-5 * max(1, d["variance"])
    want = Genotyped(cov, math.pow(10, -20.0 / 10), "haploid")
    for i, site in enumerate(j["Sites"]):
        w = want.site(i)
        assert site["GT"] == w["GT"] and site["DP"] == w["DP"] and site["ALS"] == w["ALS"] and site["HAPG"] == w["HAPG"], (i, site, w)
        assert np.allclose(site["COV"][0], w["COV"][0], rtol=0, atol=1e-9)
        assert abs(site["GT_CONF"][0] - w["GT_CONF"][0]) <= 1e-9 and site["FT"] == w["FT"]


@pytest.mark.gpu
def test_cli_files_equal_python_mirror_dumps(tmp_path):
    """Two reads files, 5000-per-batch seeding across files, non-ACGT reads: the files `gram` writes are
    byte-identical to the dumps of the Python mirror fed with the same reads."""
    from gramtools_amd import Index, quasimap_reads, dump_allele_sum, dump_allele_base, dump_grouped_allele_counts
    from gramtools_amd.synth import random_ref, snp_prg, simulate_snp_reads
    ref = random_ref(3000, 4)
    prg, pos, alts, n_alts = snp_prg(ref, 40, 5, multi_allelic_frac=0.3)
    (tmp_path / "prg").write_bytes(ints_to_prg_bytes(prg))
    reads = simulate_snp_reads(ref, pos, alts, n_alts, 5200, 60, 6)
    as_txt = ["".join("ACGT"[b - 1] for b in r) for r in reads]
    as_txt[17] = as_txt[17][:10] + "N" + as_txt[17][11:]
    f1, f2 = as_txt[:5100], as_txt[5100:]
    _write_fastq(tmp_path / "a.fq", f1)
    _write_fastq(tmp_path / "b.fq", f2)
    out = tmp_path / "run"
    r = run("genotype", "--gram_dir", str(tmp_path), "--reads", str(tmp_path / "a.fq"), str(tmp_path / "b.fq"),
            "--sample_id", "s", "--ploidy", "diploid", "--kmer_size", "6", "--genotype_dir", str(out), "--seed", "1234")
    assert r.returncode == 0, r.stdout + r.stderr
    cov = quasimap_reads(Index(prg, 6), [f1, f2], seed=1234)
    assert (out / "coverage" / "allele_sum_coverage").read_text() == dump_allele_sum(cov)
    assert (out / "coverage" / "allele_base_coverage.json").read_text() == dump_allele_base(cov)
    assert (out / "coverage" / "grouped_allele_counts_coverage.json").read_text() == dump_grouped_allele_counts(cov)
    assert f"Count skipped reads with no sequence: 2" in r.stdout
    d = cov.depth_stats()
    rs = json.loads((out / "read_stats.json").read_text())
    assert abs(rs["Read_depth"]["Mean"] - d["mean"]) < 1e-4 * max(1, d["mean"])
    assert rs["Read_depth"]["num_sites_total"] == d["num_sites_total"]


def _parse_check(tmp_path, text, threads=4, name="r.fastq", binary=None):
    import subprocess
    from gramtools_amd.build import build_gram
    path = tmp_path / name
    path.write_bytes(binary if binary is not None else text.encode())
    out = subprocess.run([build_gram(), "_parse_check", str(path), str(threads)], stdout=subprocess.PIPE, text=True)
    assert out.returncode == 0
    return out.stdout.strip().splitlines()


def test_parallel_fastq_parser_agrees_with_the_sequential_reader(tmp_path):
    """The multi-threaded four-line FASTQ fast path of the `gram` executable against its general reader: quality
    lines starting with '@' or '+', reads with N (dropped as a whole, kept as empty reads), CRLF line ends, a last
    record without newline, and more threads than records."""
    import numpy as np
    rng = np.random.default_rng(3)
    recs = []
    for i in range(2000):
        n = int(rng.integers(1, 300))
        seq = "".join("ACGTacgtN"[int(x)] for x in rng.integers(0, 9 if i % 7 == 0 else 8, size=n))
        qual = "".join(chr(int(x)) for x in rng.integers(33, 75, size=n))
        if i % 5 == 0:
            qual = "@" + qual[1:]
        if i % 11 == 0:
            qual = "+" + qual[1:]
        recs.append(f"@read{i} extra\n{seq}\n+\n{qual}\n")
    for threads in (1, 3, 16, 64):
        lines = _parse_check(tmp_path, "".join(recs), threads)
        assert lines[0].startswith("fast ") and lines[0][5:] == lines[1][5:], lines
    crlf = "".join(recs[:50]).replace("\n", "\r\n")
    lines = _parse_check(tmp_path, crlf, 4)
    assert lines[0].startswith("fast ") and lines[0][5:] == lines[1][5:], lines
    lines = _parse_check(tmp_path, "".join(recs[:7])[:-1], 64)  # no final newline, more threads than records
    assert lines[0].startswith("fast ") and lines[0][5:] == lines[1][5:], lines


def test_parallel_fastq_parser_declines_what_it_does_not_cover(tmp_path):
    import gzip
    fasta = ">a\nACGT\n>b\nGGCC\n"
    assert _parse_check(tmp_path, fasta)[0] == "fast declined"
    multi = "@a\nACGT\nACGT\n+\nIIII\nIIII\n"  # multi-line record
    assert _parse_check(tmp_path, multi)[0] == "fast declined"
    blank = "@a\nACGT\n+\nIIII\n\n@b\nAC\n+\nII\n"
    assert _parse_check(tmp_path, blank, 1)[0] == "fast declined"
    gz = gzip.compress(fasta.encode())
    assert _parse_check(tmp_path, None, 2, name="r.fa.gz", binary=gz)[0] == "fast declined"


def test_gzip_fastq_is_inflated_in_blocks_and_parsed_in_parallel(tmp_path, monkeypatch):
    """Stream blocks far smaller than the file: records straddling block ends are carried over intact."""
    import gzip
    import numpy as np
    rng = np.random.default_rng(5)
    recs = []
    for i in range(3000):
        n = int(rng.integers(1, 200))
        seq = "".join("ACGTN"[int(x)] for x in rng.integers(0, 5 if i % 9 == 0 else 4, size=n))
        qual = "".join(chr(int(x)) for x in rng.integers(33, 75, size=n))
        recs.append(f"@r{i}\n{seq}\n+\n{'@' + qual[1:] if i % 4 == 0 else qual}\n")
    gz = gzip.compress("".join(recs).encode())
    for block in ("700", "4096", "100000"):
        monkeypatch.setenv("GMX_FASTQ_BLOCK", block)
        lines = _parse_check(tmp_path, None, 5, name="r.fastq.gz", binary=gz)
        assert lines[0].startswith("fast 3000 ") and lines[0][5:] == lines[1][5:], (block, lines)


def test_parser_emits_the_same_reads_with_and_without_avx2_and_for_reads_of_one_length(tmp_path, monkeypatch):
    """The parser threads write bit planes (what gmx_map_reads_packed_host uploads); `_parse_check` unpacks them again.
    Reads of one length take the back-to-back layout, others the offsets layout; the letters are packed with AVX2 or by
    the table (GMX_NO_AVX2=1)."""
    import numpy as np
    rng = np.random.default_rng(9)
    for uniform in (True, False):
        recs = []
        for i in range(1500):
            n = 150 if uniform else int(rng.integers(1, 260))
            seq = "".join("ACGTacgtN"[int(x)] for x in rng.integers(0, 9 if i % 13 == 0 else 8, size=n))
            recs.append(f"@r{i}\n{seq}\n+\n{'I' * n}\n")
        got = []
        for no_avx in ("", "1"):
            if no_avx:
                monkeypatch.setenv("GMX_NO_AVX2", "1")
            else:
                monkeypatch.delenv("GMX_NO_AVX2", raising=False)
            lines = _parse_check(tmp_path, "".join(recs), 7)
            assert lines[0].startswith("fast 1500 ") and lines[0][5:] == lines[1][5:], lines
            got.append(lines[0])
        assert got[0] == got[1]


def test_plain_fastq_read_and_parsed_in_one_pass_with_tiny_blocks(tmp_path, monkeypatch):
    """Plain files: every parser thread reads its slice of the block from the file and parses at once, waiting for a
    neighbour's slice where a record reaches into it. Blocks and slices far smaller than a record exercise every wait."""
    import numpy as np
    rng = np.random.default_rng(12)
    recs = []
    for i in range(1200):
        n = int(rng.integers(1, 400))
        seq = "".join("ACGTN"[int(x)] for x in rng.integers(0, 5 if i % 10 == 0 else 4, size=n))
        qual = "".join(chr(int(x)) for x in rng.integers(33, 75, size=n))
        recs.append(f"@r{i}\n{seq}\n+\n{'@' + qual[1:] if i % 3 == 0 else qual}\n")
    for block, threads in (("900", 16), ("3000", 64), ("65536", 5)):
        monkeypatch.setenv("GMX_FASTQ_BLOCK", block)
        lines = _parse_check(tmp_path, "".join(recs), threads)
        assert lines[0].startswith("fast 1200 ") and lines[0][5:] == lines[1][5:], (block, lines)


@pytest.mark.gpu
def test_debug_switch_writes_the_genotyping_debug_file(tmp_path):
    """`--debug` (a global switch, main.cpp:58; after the command as the front-end passes it, genotype.py:71-93):
    site_gtyping_debug_info.txt (parameters.cpp:98) with one line per site in genotyping order (runner.cpp:66-75)."""
    case = _it_cases()[1]
    (tmp_path / "prg").write_bytes(ints_to_prg_bytes(case["prg"]["ints"]))
    reads = [op for op in case["ops"] if op["op"] == "map_reads"][0]["reads"]
    _write_fastq(tmp_path / "r.fastq", reads)
    for where in ("after", "before"):
        out = tmp_path / f"run_{where}"
        args = ["genotype", "--gram_dir", str(tmp_path), "--reads", str(tmp_path / "r.fastq"), "--sample_id", "t", "--ploidy", "haploid",
                "--kmer_size", "5", "--genotype_dir", str(out), "--max_threads", "1", "--seed", "42"]
        r = run(*(args + ["--debug"] if where == "after" else ["--debug"] + args))
        assert r.returncode == 0, r.stdout + r.stderr
        lines = (out / "site_gtyping_debug_info.txt").read_text().splitlines()
        n_sites = len(json.loads((out / "genotype" / "genotyped.json").read_text())["Sites"])
        assert len(lines) == n_sites and all(l.startswith("site index: \t") for l in lines)
        assert all(("null gt" in l) or ("next_best_seq: " in l and "next_best_cov: " in l) for l in lines)


# ---- gzip containers (gmx_gzsource.h; the reference's reader takes gzip transparently, seqread.hpp:94-180) ----------------
def _bgzf(data: bytes, block=65280, level=6, eof=True) -> bytes:
    """BGZF as bgzip writes it (SAM spec §4.1): gzip members of <= 64 KB with a BC extra field holding the member size."""
    import struct
    import zlib
    out = bytearray()
    pieces = [data[i:i + block] for i in range(0, len(data), block)] + ([b""] if eof else [])
    for piece in pieces:
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        comp = c.compress(piece) + c.flush()
        bsize = 12 + 6 + len(comp) + 8
        out += b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1)
        out += comp + struct.pack("<II", zlib.crc32(piece) & 0xFFFFFFFF, len(piece))
    return bytes(out)


def _fastq_text(n, seed, crlf=False):
    import numpy as np
    rng = np.random.default_rng(seed)
    recs = []
    for i in range(n):
        ln = int(rng.integers(1, 260))
        seq = "".join("ACGTN"[int(x)] for x in rng.integers(0, 5 if i % 9 == 0 else 4, size=ln))
        qual = "".join(chr(int(x)) for x in rng.integers(33, 75, size=ln))
        recs.append(f"@r{i}\n{seq}\n+\n{'@' + qual[1:] if i % 4 == 0 else qual}\n")
    text = "".join(recs)
    return text.replace("\n", "\r\n") if crlf else text


def _same_as_plain(tmp_path, text, binary, threads=6, name="r.fastq.gz"):
    want = _parse_check(tmp_path, text, threads, name="plain.fastq")
    got = _parse_check(tmp_path, None, threads, name=name, binary=binary)
    assert want[0].startswith("fast ") and got[0].startswith("fast ") and got[0] == want[0] and got[1][5:] == got[0][5:], (want, got)


@pytest.mark.parametrize("block", ["700", "70000", "100000000"])
def test_bgzf_members_are_inflated_side_by_side(tmp_path, monkeypatch, block):
    """BGZF: the member table is walked without inflating, members inflate in parallel into place. Tiny feed blocks force
    members through the carry buffer (a member larger than the request), large ones the batch path."""
    monkeypatch.setenv("GMX_FASTQ_BLOCK", block)
    text = _fastq_text(4000, 21)
    _same_as_plain(tmp_path, text, _bgzf(text.encode(), block=3000))
    _same_as_plain(tmp_path, text, _bgzf(text.encode()), threads=64)          # full-size members, more threads than members
    _same_as_plain(tmp_path, text, _bgzf(text.encode(), eof=False))           # no EOF marker member


def test_gzip_members_in_any_mix(tmp_path, monkeypatch):
    """cat a.gz b.gz, a BGZF run behind a plain member and the other way round, CRLF text, trailing zero bytes."""
    import gzip
    a, b, c = _fastq_text(700, 1), _fastq_text(900, 2), _fastq_text(500, 3, crlf=True)
    for block in ("900", "100000000"):
        monkeypatch.setenv("GMX_FASTQ_BLOCK", block)
        _same_as_plain(tmp_path, a + b, gzip.compress(a.encode()) + gzip.compress(b.encode()))
        _same_as_plain(tmp_path, a + b + c, gzip.compress(a.encode()) + _bgzf(b.encode(), block=5000, eof=False) + gzip.compress(c.encode()))
        _same_as_plain(tmp_path, b + a, _bgzf(b.encode(), block=4000) + gzip.compress(a.encode(), 1))
        _same_as_plain(tmp_path, c, gzip.compress(c.encode()) + b"\0" * 512)


@pytest.mark.parametrize("kind", ["plain", "bgzf", "bgzf-crc", "garbage-mid"])
def test_damaged_gzip_is_fatal(tmp_path, kind):
    """A truncated or damaged gzip file must not pass for the end of the reads (quasimap would silently report coverage of a
    part of the sample)."""
    import gzip
    import subprocess
    from gramtools_amd.build import build_gram
    text = _fastq_text(3000, 4).encode()
    if kind == "plain":
        data = gzip.compress(text)[:-2000]
    elif kind == "bgzf":
        data = _bgzf(text, block=20000)[:-3000]
    elif kind == "bgzf-crc":
        d = bytearray(_bgzf(text, block=20000))
        d[len(d) // 2] ^= 0x55
        data = bytes(d)
    else:
        d = bytearray(gzip.compress(text))
        d[len(d) // 2:len(d) // 2 + 64] = b"\xff" * 64
        data = bytes(d)
    path = tmp_path / "bad.fastq.gz"
    path.write_bytes(data)
    out = subprocess.run([build_gram(), "_parse_check", str(path), "4"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert out.returncode != 0, out.stdout


# ---- a plain gzip stream on all threads (gmx_pargz.h: block starts found by speculation, unknown windows as placeholders) ---
def _gz(data: bytes, level=6, strategy=None) -> bytes:
    import zlib
    c = zlib.compressobj(level, zlib.DEFLATED, 31, 8, zlib.Z_DEFAULT_STRATEGY if strategy is None else strategy)
    return c.compress(data) + c.flush()


@pytest.mark.parametrize("chunk,threads", [("20000", 4), ("60000", 8), ("7000", 3), ("300000", 16)])
def test_plain_gzip_decoded_on_all_threads(tmp_path, monkeypatch, chunk, threads):
    """Pieces of the compressed stream decoded side by side: every piece must start exactly where the one before it ends,
    placeholders of the unknown 32 KB are resolved in stream order, CRC-32 and length checked against the trailer. The
    parsed reads equal those of the plain file, whatever the piece size and the thread count."""
    monkeypatch.setenv("GMX_PARGZ_MIN", "1000")
    monkeypatch.setenv("GMX_PARGZ_CHUNK", chunk)
    text = _fastq_text(9000, 31)
    for level in (1, 6, 9):
        _same_as_plain(tmp_path, text, _gz(text.encode(), level), threads=threads)


def test_parallel_gzip_really_ran_and_falls_back_where_it_cannot(tmp_path, monkeypatch):
    """`gram _gz_info` reports how a file was decompressed. Dynamic-Huffman streams are taken apart into pieces; streams with
    no dynamic block to find (fixed codes, stored blocks) go to zlib from the last verified bit, primed with the known
    window; concatenated members each get their own treatment."""
    import subprocess
    from gramtools_amd.build import build_gram
    monkeypatch.setenv("GMX_PARGZ_MIN", "1000")
    monkeypatch.setenv("GMX_PARGZ_CHUNK", "30000")
    import zlib
    text = _fastq_text(9000, 33)

    def info(binary, name="r.fastq.gz"):
        (tmp_path / name).write_bytes(binary)
        out = subprocess.run([build_gram(), "_gz_info", str(tmp_path / name), "6"], stdout=subprocess.PIPE, text=True)
        assert out.returncode == 0, out.stdout
        return dict(kv.split("=") for kv in out.stdout.split())

    a = info(_gz(text.encode()))
    assert int(a["bytes"]) == len(text.encode()) and int(a["pieces"]) > 8 and int(a["crc"]) == zlib.crc32(text.encode())
    for strategy, level in ((zlib.Z_FIXED, 6), (None, 0)):
        b = info(_gz(text.encode(), level, strategy))
        assert int(b["bytes"]) == len(text.encode()) and int(b["crc"]) == zlib.crc32(text.encode())
        assert int(b["stream_bytes"]) > len(text) // 2   # most of it through zlib
        _same_as_plain(tmp_path, text, _gz(text.encode(), level, strategy))
    more = _fastq_text(2500, 34)
    two = _gz(text.encode()) + _gz(more.encode(), 9)
    c = info(two)
    assert int(c["bytes"]) == len(text.encode()) + len(more.encode()) and int(c["pieces"]) > int(a["pieces"])
    _same_as_plain(tmp_path, text + more, two)


@pytest.mark.parametrize("kind", ["truncated", "flipped-bit", "bad-crc", "bad-length"])
def test_damaged_plain_gzip_is_fatal_with_the_parallel_decoder(tmp_path, monkeypatch, kind):
    import subprocess
    from gramtools_amd.build import build_gram
    monkeypatch.setenv("GMX_PARGZ_MIN", "1000")
    monkeypatch.setenv("GMX_PARGZ_CHUNK", "25000")
    d = bytearray(_gz(_fastq_text(8000, 35).encode()))
    if kind == "truncated":
        d = d[:len(d) * 2 // 3]
    elif kind == "flipped-bit":
        d[len(d) // 2] ^= 0x10
    elif kind == "bad-crc":
        d[-6] ^= 0xFF
    else:
        d[-2] ^= 0x01
    path = tmp_path / "bad.fastq.gz"
    path.write_bytes(bytes(d))
    out = subprocess.run([build_gram(), "_parse_check", str(path), "5"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert out.returncode != 0, out.stdout


def test_parallel_gzip_bounds_its_pieces_and_rejects_distances_before_the_member(tmp_path, monkeypatch):
    """(round 5, ADVICE) (a) A stream that inflates more than 32-fold per piece — one read repeated — is not damage: the
    parallel decoder stops at its per-piece bound and zlib takes the member over from the last verified bit; the reads are
    those of the plain file. (b) A back-reference that reaches before the first byte of a member is rejected as zlib
    rejects it ("invalid distance too far back"), not resolved to zero bytes."""
    import subprocess
    import zlib
    from gramtools_amd.build import build_gram
    monkeypatch.setenv("GMX_PARGZ_MIN", "1000")
    monkeypatch.setenv("GMX_PARGZ_CHUNK", "4000")
    one = _fastq_text(1, 41)
    text = one * 30000          # ~10 MB that deflate packs ~300-fold: far beyond 32 x 4000 bytes per piece
    binary = _gz(text.encode(), 9)
    assert len(text) > 100 * len(binary)
    (tmp_path / "rep.fastq.gz").write_bytes(binary)
    out = subprocess.run([build_gram(), "_gz_info", str(tmp_path / "rep.fastq.gz"), "6"], stdout=subprocess.PIPE, text=True)
    assert out.returncode == 0, out.stdout
    info = dict(kv.split("=") for kv in out.stdout.split())
    assert int(info["bytes"]) == len(text.encode()) and int(info["crc"]) == zlib.crc32(text.encode())
    assert int(info["stream_bytes"]) > len(text) // 2   # zlib did the work
    _same_as_plain(tmp_path, text, binary)
    # (b) a raw deflate stream whose first block copies 10 bytes from distance 5 with nothing in front: fixed-Huffman block,
    # length code 264 (len 10), distance code 4 (dist 5): built bit by bit
    bits = []

    def put(value, n, msb_first=False):
        for i in (range(n - 1, -1, -1) if msb_first else range(n)):
            bits.append((value >> i) & 1)
    put(0, 1)            # BFINAL = 0
    put(1, 2)            # BTYPE = 01 fixed
    put(0b0001000, 7, True)   # length symbol 264 (7-bit code 0001000): length 10
    put(0b00100, 5, True)     # distance symbol 4: distance 5, one extra bit
    put(0, 1)
    put(0, 7, True)      # end of block (256)
    put(0, 1)            # an empty stored block: BFINAL = 0, BTYPE = 00, padding to the byte, LEN = 0, NLEN = 0xFFFF
    put(0, 2)
    while len(bits) % 8:
        bits.append(0)
    raw = bytes(sum(b << i for i, b in enumerate(bits[j:j + 8])) for j in range(0, len(bits), 8)) + b"\x00\x00\xff\xff"
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    raw += c.compress(_fastq_text(6000, 42).encode()) + c.flush()   # the rest of the member: ordinary deflate blocks, > 100 KB
    with pytest.raises(zlib.error, match="too far back"):
        zlib.decompress(raw, -15)
    member = b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03" + raw + (0).to_bytes(4, "little") + (10).to_bytes(4, "little")
    assert len(member) > 50_000   # large enough for the parallel decoder (GMX_PARGZ_MIN = 1000): ITS first piece must refuse
    (tmp_path / "far.fastq.gz").write_bytes(member)
    out = subprocess.run([build_gram(), "_parse_check", str(tmp_path / "far.fastq.gz"), "4"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert out.returncode != 0, out.stdout


@pytest.mark.gpu
def test_many_samples_in_one_call_equal_one_call_each(tmp_path):
    """`gram genotype --samples_list` (round 5: one index load and upload for many samples of one species — the reference's
    stated use, README.md:156): every file of every sample is byte-identical to that of a call of its own with the same
    --seed (files written: genotype/parameters.cpp:94-110)."""
    import numpy as np
    from gramtools_amd.synth import random_ref, snp_prg, simulate_snp_reads
    ref = random_ref(60_000, 91)
    prg, pos, alts, n_alts = snp_prg(ref, 900, 92, multi_allelic_frac=0.1)
    (tmp_path / "prg").write_bytes(ints_to_prg_bytes([int(x) for x in prg]))
    assert run("build", "--gram_dir", str(tmp_path), "--kmer_size", "7", "--max_threads", "4").returncode == 0
    letters = np.frombuffer(b"NACGT", dtype=np.uint8)
    lines = []
    for s in range(3):
        paths = []
        for f in range(1 + s % 2):  # samples with one and with two reads files (5 000-draw rule per file)
            rd = simulate_snp_reads(ref, pos, alts, n_alts, 6000 + 701 * s + 13 * f, 120 + 10 * s, 93 + 10 * s + f)
            fq = tmp_path / f"s{s}_{f}.fastq"
            _write_fastq(fq, [letters[r].tobytes().decode() for r in rd], False)
            paths.append(str(fq))
        lines.append((f"smp{s}", paths))
    common_args = ["--gram_dir", str(tmp_path), "--ploidy", "haploid", "--kmer_size", "7", "--max_threads", "4", "--seed", "7"]
    for sid, paths in lines:
        r = run("genotype", *common_args, "--reads", *paths, "--sample_id", sid, "--genotype_dir", str(tmp_path / f"single_{sid}"))
        assert r.returncode == 0, r.stdout + r.stderr
    (tmp_path / "samples.tsv").write_text("".join("\t".join([sid, str(tmp_path / f"multi_{sid}")] + paths) + "\n" for sid, paths in lines))
    r = run("genotype", *common_args, "--samples_list", str(tmp_path / "samples.tsv"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("Count all reads:") == 3
    names = ["coverage/allele_sum_coverage", "coverage/allele_base_coverage.json", "coverage/grouped_allele_counts_coverage.json",
             "read_stats.json", "genotype/genotyped.json", "genotype/genotyped.vcf.gz", "genotype/personalised_reference.fasta"]
    for sid, _ in lines:
        for n in names:
            a, b = (tmp_path / f"single_{sid}" / n).read_bytes(), (tmp_path / f"multi_{sid}" / n).read_bytes()
            assert a == b and len(a) > 0, (sid, n)
    # and the list replaces the single-sample options
    bad = run("genotype", *common_args, "--samples_list", str(tmp_path / "samples.tsv"), "--sample_id", "x")
    assert bad.returncode != 0
