"""CPU-side checks: on-disk formats (oracle writer vs library writer/reader), the C-ABI surface of
libr3dm.so (every symbol include/*.h declares is exported; no compute without a GPU), host-only
graph utilities, and the "fails loudly without a GPU" rule.  No GPU needed.
"""
import ctypes
import os
import re
import struct

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PAIRS = np.array([[0, 1], [0, 3], [2, 5]], np.uint32)
COUNTS = np.array([2, 0, 3], np.uint32)
MATCHES = np.array([[4, 7], [9, 1], [0, 0], [5, 6], [8, 2]], np.uint32)

GOLDEN_TXT = "0 1\n2\n4 7\n9 1\n2 5\n3\n0 0\n5 6\n8 2\n"
# cereal PortableBinaryOutputArchive of std::map<pair<u32,u32>, vector<IndMatch>>:
# endianness flag, u64 map size, then (u32 I, u32 J, u64 count, count x (u32 i, u32 j)) per entry
GOLDEN_BIN = (b"\x01" + struct.pack("<Q", 2)
              + struct.pack("<IIQ", 0, 1, 2) + struct.pack("<IIII", 4, 7, 9, 1)
              + struct.pack("<IIQ", 2, 5, 3) + struct.pack("<IIIIII", 0, 0, 5, 6, 8, 2))


def test_oracle_txt_bin_golden(oracle, tmp_path):
    t = str(tmp_path / "matches.putative.txt"); b = str(tmp_path / "matches.putative.bin")
    oracle.save_matches(t, PAIRS, COUNTS, MATCHES)
    oracle.save_matches(b, PAIRS, COUNTS, MATCHES)
    assert open(t).read() == GOLDEN_TXT                      # empty pair (0,3) never enters the map
    assert open(b, "rb").read() == GOLDEN_BIN
    for path in (t, b):
        p, c, m = oracle.load_matches(path)
        assert p.tolist() == [[0, 1], [2, 5]] and c.tolist() == [2, 3] and np.array_equal(m, MATCHES)


def test_library_writer_reader_match_the_goldens(tmp_path):
    from regard3d_amd import api
    offs = np.concatenate([[0], np.cumsum(COUNTS)]).astype(np.uint64)
    g = api.Graph.from_csr(PAIRS, offs, MATCHES)
    assert g.num_pairs == 2 and g.num_matches == 5            # from_csr drops the empty entry
    t = str(tmp_path / "matches.f.txt"); b = str(tmp_path / "matches.f.bin")
    g.save(t); g.save(b)
    assert open(t).read() == GOLDEN_TXT
    assert open(b, "rb").read() == GOLDEN_BIN
    for path in (t, b):
        g2 = api.Graph.load(path)
        assert g2.pairs.tolist() == [[0, 1], [2, 5]] and np.array_equal(g2.matches, MATCHES)
    with pytest.raises(api.R3dmError):
        g.save(str(tmp_path / "matches.xyz"))                 # Save() dispatches on the extension
    with pytest.raises(api.R3dmError):
        api.Graph.load(str(tmp_path / "missing.txt"))


def test_large_match_files_are_the_oracle_writers_bytes(oracle, tmp_path):
    """r3dm_save_matches formats a long .txt on several host threads (runs of pairs of about equal match counts, each into its own
    buffer, written out in order): the bytes are those of the restatement's writer -- 300 k matches over 700 pairs, some of them
    empty, numbers of one to ten digits -- and the file reads back to the same graph"""
    from regard3d_amd import api
    rng = np.random.default_rng(11)
    P = 700
    pairs = np.array([(i, j) for i in range(60) for j in range(i + 1, 60)][:P], np.uint32)
    counts = rng.integers(0, 900, P).astype(np.uint32)
    counts[[3, 4, 250, 699]] = 0
    counts[17] = 9000
    n = int(counts.sum())
    mag = rng.integers(0, 10, n)                                                   # digits per number, uniformly
    matches = np.stack([(rng.random(n) * 10.0 ** mag).astype(np.uint64) % (1 << 32), rng.integers(0, 1 << 32, n, dtype=np.uint64)], 1).astype(np.uint32)
    g = api.Graph.from_csr(pairs, np.r_[0, np.cumsum(counts.astype(np.uint64))].astype(np.uint64), matches)
    t_lib, t_orc = str(tmp_path / "lib.txt"), str(tmp_path / "orc.txt")
    g.save(t_lib)
    keep = counts > 0                                                              # (a graph holds no empty pairs: from_csr drops them)
    oracle.save_matches(t_orc, pairs[keep], counts[keep], matches)
    assert n > 250000 and open(t_lib, "rb").read() == open(t_orc, "rb").read()
    back = api.Graph.load(t_lib)
    assert np.array_equal(back.pairs, g.pairs) and np.array_equal(back.offsets, g.offsets) and np.array_equal(back.matches, g.matches)


def test_match_files_longer_than_one_writer_round(oracle, tmp_path):
    """the .txt writer formats at most ~64 MiB of text at a time (rounds of pairs; a graph of 1e8 matches must not be held as text):
    7 M matches -- three rounds, one of them a single pair longer than a round -- give the restatement's bytes"""
    from regard3d_amd import api
    rng = np.random.default_rng(12)
    counts = np.array([1500000, 200000, 3300000, 5, 900000, 1100000], np.uint32)
    pairs = np.array([(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)], np.uint32)
    n = int(counts.sum())
    matches = rng.integers(0, 1 << 20, (n, 2), dtype=np.uint64).astype(np.uint32)
    g = api.Graph.from_csr(pairs, np.r_[0, np.cumsum(counts.astype(np.uint64))].astype(np.uint64), matches)
    t_lib, t_orc = str(tmp_path / "lib.txt"), str(tmp_path / "orc.txt")
    g.save(t_lib)
    oracle.save_matches(t_orc, pairs, counts, matches)
    import filecmp
    assert filecmp.cmp(t_lib, t_orc, shallow=False)


def test_graph_from_csr_orders_pairs_and_merge():
    from regard3d_amd import api
    a = api.Graph.from_csr(np.array([[2, 5], [0, 1]], np.uint32), np.array([0, 3, 5], np.uint64),
                           np.array([[0, 0], [5, 6], [8, 2], [4, 7], [9, 1]], np.uint32))
    assert a.pairs.tolist() == [[0, 1], [2, 5]] and a.matches.tolist() == MATCHES.tolist()
    b = api.Graph.from_csr(np.array([[1, 4]], np.uint32), np.array([0, 1], np.uint64), np.array([[3, 3]], np.uint32))
    m = api.Graph.merge([b, a])
    assert m.pairs.tolist() == [[0, 1], [1, 4], [2, 5]]
    assert m.as_dict()[(1, 4)].tolist() == [[3, 3]]
    e = api.Graph.merge([])
    assert e.num_pairs == 0 and e.num_matches == 0


def test_desc_and_feat_formats(oracle, tmp_path):
    rng = np.random.default_rng(0)
    d = rng.normal(size=(5, 144)).astype(np.float32)
    p = str(tmp_path / "img.desc")
    assert oracle.lib().orc_save_desc(p.encode(), ctypes.c_uint64(5), ctypes.c_size_t(144 * 4), d.ctypes.data_as(ctypes.c_void_p)) == 0
    raw = open(p, "rb").read()
    assert len(raw) == 8 + 5 * 144 * 4 and struct.unpack("<Q", raw[:8])[0] == 5      # 8-byte count header
    assert np.array_equal(np.frombuffer(raw[8:], np.float32).reshape(5, 144), d)
    f = str(tmp_path / "img.feat")
    xyso = np.array([[1.5, 2.25, 3.0, 0.5], [100.125, 7.0, 1.0, -1.0]], np.float32)
    assert oracle.lib().orc_save_feat(f.encode(), 2, xyso.ctypes.data_as(ctypes.c_void_p)) == 0
    assert open(f).read() == "1.5 2.25 3 0.5\n100.125 7 1 -1\n"


def _declared_symbols():
    names = set()
    for hdr in os.listdir(os.path.join(ROOT, "include")):
        txt = open(os.path.join(ROOT, "include", hdr)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        txt = re.sub(r"//.*", "", txt)
        names |= set(re.findall(r"\b(r3dm_[A-Za-z0-9_]+)\s*\(", txt))
    return names


def test_library_exports_every_declared_symbol():
    from regard3d_amd import api
    L = api.load_library()
    declared = _declared_symbols()
    assert len(declared) >= 20
    missing = [n for n in sorted(declared) if not hasattr(L, n)]
    assert not missing, f"libr3dm.so lacks: {missing}"
    assert set(api.EXPORTS) <= declared


def test_no_cpu_fallback_without_gpu():
    """r3dm_create must fail (not silently degrade) when no gfx950 GPU is visible."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from regard3d_amd import api
    with pytest.raises(api.R3dmError):
        api.Context(0)


def test_registration_entries_reject_null_handles_without_a_gpu():
    """the registration entries of round 6 validate their arguments before they touch the device: a null context (what a host holds
    after a failed r3dm_create) is R3DM_ERR_INVALID = 1, never a crash -- callable without a GPU"""
    import ctypes as C
    from regard3d_amd import api
    L = api.load_library()
    view = api.ViewDesc(0, 0, 0, 0, 128, api.F32, None, None)
    assert L.r3dm_set_images(None, C.byref(view), 1) != 0
    assert L.r3dm_images_wait(None) != 0
    assert L.r3dm_view_info(None, 0, None, None, None, None) != 0
    assert L.r3dm_memory_info(None, None, None, None) != 0
    L.r3dm_set_background_nice.argtypes = [C.c_void_p, C.c_int]
    assert L.r3dm_set_background_nice(None, 10) != 0


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "regard3d_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(dirpath, fn)).read()
                assert "pyoracle" not in txt and "liboracle" not in txt and "r3d_oracle.h" not in txt, fn


def test_product_library_never_reads_the_environment():
    """The A/B kernel variants (two of them ablations that return garbage), traces and test hooks live in the developer build
    only (build.sh dev -> libr3dm_dev.so, -DR3DM_DEVTOOLS): the product sources contain no getenv call outside the
    #ifdef'd declaration and the product library does not import the symbol."""
    import subprocess
    csrc = os.path.join(ROOT, "regard3d_amd", "csrc")
    for fn in os.listdir(csrc):
        txt = open(os.path.join(csrc, fn)).read()
        txt = re.sub(r"//.*", "", txt)
        assert "getenv" not in txt, fn
    so = os.path.join(ROOT, "regard3d_amd", "libr3dm.so")
    syms = subprocess.run(["nm", "-D", so], capture_output=True, text=True).stdout
    assert "getenv" not in syms
    # the timing-only ablation kernels (PIPE == 9, ABL == 1) are not even compiled into the product library
    names = subprocess.run(["nm", "-C", so], capture_output=True, text=True).stdout
    assert "l2_knn2_mfma_kernel<16, 2, 4, 9, 2>" not in names and "l2_knn2_int_kernel<8, 2, 4, 2, 1>" not in names
    assert "l2_knn2_mfma_kernel<16, 2, 4, 3, 2>" in names


def test_shard_pairs_is_balanced_keeps_rows_together_and_is_a_partition():
    from regard3d_amd import api, dist
    for n, world in ((200, 8), (1000, 8), (37, 2), (10, 4), (5, 8)):
        ii, jj = np.triu_indices(n, k=1)
        pairs = np.stack([ii, jj], 1).astype(np.uint32)
        owner = api.shard_owner(pairs, world)
        # rows of I stay on one rank
        for I in range(n - 1):
            assert len(set(owner[pairs[:, 0] == I].tolist())) == 1
        counts = np.bincount(owner, minlength=world)
        if n >= 4 * world:
            assert counts.max() - counts.min() <= n, (n, world, counts)
        parts = [dist.shard_pairs(pairs, r, world) for r in range(world)]
        allp = np.concatenate(parts)
        assert allp.shape == pairs.shape
        assert np.array_equal(allp[np.lexsort((allp[:, 1], allp[:, 0]))], pairs)
    # the rule itself: rows by decreasing pair count (ascending I among equals), dealt 0..W-1, W-1..0, ...
    pairs = np.array([[0, 1], [0, 2], [0, 3], [1, 2], [1, 3], [2, 3], [7, 9], [7, 8]], np.uint32)
    assert api.shard_owner(pairs, 2).tolist() == [0, 0, 0, 1, 1, 0, 1, 1]
    assert api.shard_owner(pairs, 1).tolist() == [0] * 8
    assert api.shard_owner(np.zeros((0, 2), np.uint32), 3).size == 0


def test_corrupt_match_and_descriptor_files_fail_cleanly(tmp_path):
    """counts read from disk must not size allocations (std::bad_alloc through extern "C" = std::terminate in the host)"""
    from regard3d_amd import api
    good = tmp_path / "g.bin"
    api.Graph.from_csr(PAIRS, np.concatenate([[0], np.cumsum(COUNTS)]).astype(np.uint64), MATCHES).save(str(good))
    raw = open(good, "rb").read()
    # a pair count / a match count far beyond the file size, and a truncated file
    bad1 = tmp_path / "bad1.bin"; bad1.write_bytes(raw[:1] + struct.pack("<Q", 1 << 40) + raw[9:])
    bad2 = tmp_path / "bad2.bin"; bad2.write_bytes(raw[:17] + struct.pack("<Q", (1 << 32) - 1) + raw[25:])
    bad3 = tmp_path / "bad3.bin"; bad3.write_bytes(raw[:-5])
    bad4 = tmp_path / "bad4.txt"; bad4.write_text("0 1\n99999999999\n0 0\n")
    for p in (bad1, bad2, bad3, bad4):
        with pytest.raises(api.R3dmError):
            api.Graph.load(str(p))
    assert api.Graph.load(str(good)).num_matches == len(MATCHES)


def test_host_thread_budget_respects_the_cores_the_process_owns():
    """r3dm_host_threads (include/r3dm.h): at least 1, at most what was asked for, and never more than the affinity mask / the cgroup
    CPU quota leave; bench.py's host_cores() reads the same sources"""
    import ctypes as C
    from regard3d_amd import api
    L = api.load_library()
    L.r3dm_host_threads.argtypes = [C.c_int]; L.r3dm_host_threads.restype = C.c_int
    assert L.r3dm_host_threads(1) == 1 and L.r3dm_host_threads(0) == 1
    t = L.r3dm_host_threads(64)
    assert 1 <= t <= 64 and t <= len(os.sched_getaffinity(0))
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    cores, seen = bench.host_cores()
    assert 1 <= cores <= seen and t <= max(1, cores)


def test_kernel_fingerprints_of_the_built_library():
    """bench.py keys profiles/pmc_traffic.json to the machine code of the kernel that was profiled (regard3d_amd/codeobj.py): the
    built library must expose the kernels the entries name, a fingerprint must not depend on anything but the kernel's bytes
    (two reads agree; another kernel differs), and an entry carries either a fingerprint or the older source hash"""
    import json
    from regard3d_amd.codeobj import kernel_code_hashes, kernel_hash, mangled_needle
    lib = os.path.join(ROOT, "regard3d_amd", "libr3dm.so")
    hs = kernel_code_hashes(lib)
    assert len(hs) > 50 and any("l2_knn2_mfma_kernel" in k for k in hs) and any("acransac_coop_kernel" in k for k in hs)
    n = mangled_needle("l2_knn2_mfma_kernel<16, 2, 4, 3, 2>")
    assert n == "l2_knn2_mfma_kernelILi16ELi2ELi4ELi3ELi2EE"
    a, b = kernel_hash(lib, n), kernel_hash(lib, n)
    assert a is not None and a == b and a != kernel_hash(lib, mangled_needle("l2_knn2_mfma_kernel<18, 2, 3, 3, 2>"))
    assert kernel_hash(lib, "no_such_kernel") is None
    t = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    for key, ent in t.items():
        if key.startswith("_"):
            continue
        assert ent.get("code_sha16") or ent.get("source_sha16"), key
        if ent.get("code_sha16"):
            assert kernel_hash(lib, mangled_needle(ent["kernel"])) is not None, key      # the kernel exists in today's library
