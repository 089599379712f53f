"""CPU (-m "not gpu") tests of the host side: the C-ABI library loads and exports every symbol
include/drfe.h declares, fails loudly without a GPU, the synthetic generator is deterministic, and the
multi-rank plumbing works over gloo with world_size 2."""
import os
import re
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    """every function include/*.h declares: the drop-in boundary (drfe.h) and the test hooks (drfe_debug.h)"""
    names = set()
    for h in ("drfe.h", "drfe_debug.h"):
        txt = open(os.path.join(ROOT, "include", h)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        names |= set(re.findall(r"\b(drfe_[a-z0-9_]+)\s*\(", txt))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    from dr_slam_amd import lib
    L = lib.load()
    names = _header_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), f"libdrfe.so does not export {n}"
    assert sorted(lib.SYMBOLS) == names
    assert b"gfx950" in L.drfe_version()


def test_boundary_header_has_no_test_hooks():
    txt = open(os.path.join(ROOT, "include", "drfe.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    assert not re.findall(r"\bdrfe_debug_[a-z0-9_]+\s*\(", txt)


def test_struct_layouts_match_header():
    import ctypes as C
    from dr_slam_amd import lib
    assert lib.KP_DTYPE.itemsize == 28 and lib.MAPPOINT_DTYPE.itemsize == 48 and lib.TRACKED_DTYPE.itemsize == 56
    assert C.sizeof(lib.Config) == 36 and C.sizeof(lib.Camera) == 40


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from dr_slam_amd import lib
    with pytest.raises(lib.DrfeError, match="no CPU path|HIP"):
        lib.Context()


def test_product_never_imports_oracle():
    """The product package must not reference the oracle (no CPU fallback)."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "dr_slam_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "from oracle" not in src and "import oracle" not in src and "libdrfe_oracle" not in src, f


def test_synth_is_deterministic():
    from dr_slam_amd import synth
    cam = synth.TUM3.scaled(0.25)
    a = list(synth.sequence(2, 2, cam=cam))
    b = list(synth.sequence(2, 2, cam=cam))
    for (g1, d1, T1), (g2, d2, T2) in zip(a, b):
        assert np.array_equal(g1, g2) and np.array_equal(d1, d2) and np.array_equal(T1, T2)
    g, d, _ = a[0]
    assert g.dtype == np.uint8 and d.dtype == np.uint16 and g.shape == (120, 160)
    assert 0.005 < (d == 0).mean() < 0.05            # ~2 % holes
    assert (d > 5.0 * cam.depth_factor).mean() > 0.05   # a wall beyond 5 m
    # SplitMix64 reference vector (seed 0 stream: first output of the published generator)
    assert int(synth.splitmix64(np.array([0], np.uint64))[0]) == 0xE220A8397B1DCDAF


def test_pingpong_order():
    from dr_slam_amd.sharding import pingpong_order
    assert pingpong_order(10, 4) == [0, 1, 2, 3, 2, 1, 0, 1, 2, 3]
    assert pingpong_order(3, 1) == [0, 0, 0]
    o = pingpong_order(64, 8)
    assert all(abs(a - b) == 1 for a, b in zip(o, o[1:]))


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dr_slam_amd import sharding
    dev = torch.device("cpu")
    tab = np.arange(8, dtype=np.float32) * (1.0 if rank == 0 else -1.0)
    got = sharding.broadcast_tables(tab, dev, dist)
    el, n = sharding.reduce_elapsed_and_frames(1.0 + rank, 64 * (rank + 1), dev, dist)
    # the vocabulary travels as one flat blob from rank 0 (bench.py does the same over RCCL)
    from dr_slam_amd import vocabulary as V
    ref = V.make_synthetic(5, 3, seed=9)
    blob = ref.pack() if rank == 0 else np.zeros(ref.pack().size, np.uint8)
    voc = V.Vocabulary.unpack(sharding.broadcast_tables(blob, dev, dist))
    assert np.array_equal(voc.desc, ref.desc) and np.array_equal(voc.weight, ref.weight) and voc.k == 5 and voc.L == 3
    q.put((rank, got.tolist(), el, n, sharding.rank_seed(10, rank)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_sharding():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, tab, el, n, seed in res:
        assert tab == list(np.arange(8, dtype=np.float32))   # rank 0's tables everywhere
        assert el == 2.0 and n == 64 + 128                   # MAX of time, SUM of frames
        assert seed == 10 + rank                             # one independent sequence per rank


def test_eight_rank_gloo_sharding():
    """BASELINE config 4's world size (eight ranks, one sequence each) over gloo on the CPU: the one initial exchange reaches every
    rank, the report reduces MAX of time / SUM of frames over all eight, every rank draws its own sequence and intrinsics."""
    import torch.multiprocessing as mp
    from dr_slam_amd import sharding
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(8))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert [r[0] for r in res] == list(range(8))
    for rank, tab, el, n, seed in res:
        assert tab == list(np.arange(8, dtype=np.float32))
        assert el == 8.0 and n == 64 * sum(range(1, 9))
        assert seed == 10 + rank
    cams = [sharding.rank_camera(r).name for r in range(8)] if hasattr(sharding, "rank_camera") else None
    if cams:
        assert len(set(cams)) == 3                               # TUM1 / TUM2 / TUM3 cycling


def test_image_bounds_host_entry_matches_oracle(oracle_mod):
    """drfe_frame_image_bounds is pure host code (Frame::ComputeImageBounds): callable without a GPU, bit-equal to
    the oracle's restatement of cv::undistortPoints on the four corners."""
    import ctypes as C
    from dr_slam_amd import lib, synth
    L = lib.load()
    for cam in (synth.TUM1, synth.TUM2, synth.TUM3):
        c = lib.make_camera(cam.fx, cam.fy, cam.cx, cam.cy, cam.bf, cam.depth_factor, cam.w, cam.h)
        d = np.ascontiguousarray(cam.dist, np.float32)
        out = np.zeros(4, np.float32)
        rc = L.drfe_frame_image_bounds(C.byref(c), d.ctypes.data_as(C.c_void_p), len(d), cam.w, cam.h, out.ctypes.data_as(C.c_void_p))
        assert rc == 0
        K = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32)
        ref = oracle_mod.image_bounds(cam.w, cam.h, K, d if len(d) else [0.0])
        assert np.array_equal(out.view(np.uint32), ref.view(np.uint32)), (out, ref)


# ---- Frame::isLineGood (row a-8): host entry of libdrfe.so vs the numpy oracle, no GPU involved -----------------

def _lines_and_depth(oracle_mod, seed=2, kind="room_boxes"):
    from dr_slam_amd import lib, synth
    g, d16, _ = next(synth.sequence(seed, 1, kind=kind))
    o = oracle_mod.extract_lines(g)
    kl = np.zeros(len(o["lines"]), lib.KEYLINE_DTYPE)
    for a, b in (("start_point_x", "startPointX"), ("start_point_y", "startPointY"), ("end_point_x", "endPointX"),
                 ("end_point_y", "endPointY"), ("pt_x", "ptX"), ("pt_y", "ptY"), ("angle", "angle"), ("octave", "octave")):
        kl[a] = o["lines"][b]
    depth = oracle_mod.depth_to_float(d16, np.float32(1.0) / np.float32(synth.TUM3.depth_factor))
    return kl, depth


def test_glibc_rand_restatement_matches_libc():
    import ctypes
    from oracle.line3d_oracle import GlibcRand
    libc = ctypes.CDLL("libc.so.6")
    for seed in (1, 7, 20240101):
        libc.srand(seed)
        g = GlibcRand(seed)
        assert [libc.rand() for _ in range(500)] == [g() for _ in range(500)]


def test_is_line_good_as_shipped_rejects_every_line(oracle_mod):
    """mK is CV_32F but compPt3dCov reads K.at<double>(0,0): f is subnormal, z/f = inf, the covariance is NaN and no
    sample is ever an inlier — the reference leaves mvDepthLine = -1 and mvLines3D = 0; product and oracle agree."""
    from dr_slam_amd import lib, synth
    from oracle import line3d_oracle as L3
    cam = synth.TUM3
    K = np.array([cam.fx, 0, cam.cx, 0, cam.fy, cam.cy, 0, 0, 1], np.float32)
    f = L3.focal_as_reference_reads_it(K)
    with np.errstate(over="ignore"):
        assert 0 < f < 1e-300 and np.isinf(np.float64(0.5) / f)
    kl, depth = _lines_and_depth(oracle_mod)
    assert len(kl) >= 20
    inv = (np.float32(1) / np.float32(cam.fx), np.float32(1) / np.float32(cam.fy))
    dl, l3, ni, good = lib.lines_is_good(kl, depth, K, cam.cx, cam.cy, inv[0], inv[1], k_as_f64=False)
    odl, ol3, oni = L3.is_line_good(kl, depth, K, False, cam.cx, cam.cy, inv[0], inv[1])
    assert good == 0 and (dl == -1).all() and (l3 == 0).all() and (ni == 0).all()
    assert np.array_equal(dl, odl) and np.array_equal(l3, ol3) and np.array_equal(ni, oni)


def test_is_line_good_as_intended_matches_numpy_oracle(oracle_mod):
    """k_as_f64: f = fx.  Same accept decisions, inlier counts and depth; 3-D end points equal up to the A/B swap the
    singular-vector sign allows (the oracle uses LAPACK SVD, the product a symmetric eigen-solve)."""
    from dr_slam_amd import lib, synth
    from oracle import line3d_oracle as L3
    cam = synth.TUM3
    K = np.array([cam.fx, 0, cam.cx, 0, cam.fy, cam.cy, 0, 0, 1], np.float32)
    inv = (np.float32(1) / np.float32(cam.fx), np.float32(1) / np.float32(cam.fy))
    total = 0
    for seed, kind in ((2, "room_boxes"), (5, "corridor")):
        kl, depth = _lines_and_depth(oracle_mod, seed, kind)
        dl, l3, ni, good = lib.lines_is_good(kl, depth, K, cam.cx, cam.cy, inv[0], inv[1], k_as_f64=True, seed=1)
        odl, ol3, oni = L3.is_line_good(kl, depth, K, True, cam.cx, cam.cy, inv[0], inv[1], seed=1)
        assert np.array_equal(ni, oni), (ni, oni)
        assert np.array_equal(dl.view(np.uint32), odl.view(np.uint32))
        for a, b in zip(l3, ol3):
            same = np.allclose(a, b, atol=1e-9)
            swapped = np.allclose(a, np.concatenate([b[3:], b[:3]]), atol=1e-9)
            assert same or swapped
        assert good == int((dl >= 0).sum())
        total += good
    assert total >= 10          # most synthetic wall/box edges carry depth and lift to 3-D


def test_rank_workloads_and_parallel_render():
    """Config 4's partition: seeds 10..17, intrinsics cycling TUM1/TUM2/TUM3, 256 frames each;
    the pooled renderer returns exactly what synth.sequence yields."""
    from dr_slam_amd import sharding, synth
    cams = [sharding.rank_workload(4, r)[1] for r in range(8)]
    assert [sharding.rank_workload(4, r)[0] for r in range(8)] == list(range(10, 18))
    assert cams[0] is synth.TUM1 and cams[1] is synth.TUM2 and cams[2] is synth.TUM3 and cams[3] is synth.TUM1
    assert all(sharding.rank_workload(4, r)[3] == 256 for r in range(8))
    assert sharding.rank_workload(2, 0)[1] is synth.TUM3
    cam = synth.TUM1.scaled(0.25)
    a = list(synth.sequence(11, 5, cam=cam))
    b = sharding.render_sequence(11, 5, cam, workers=2)
    for (g1, d1, T1), (g2, d2, T2) in zip(a, b):
        assert np.array_equal(g1, g2) and np.array_equal(d1, d2) and np.array_equal(T1, T2)
    assert sharding.host_cpus() >= 1


def test_bench_launcher_starts_n_fresh_ranks(tmp_path):
    """`bench.py --gpus 2` without a launcher spawns 2 rank processes with RANK / WORLD_SIZE / MASTER_* set and
    relays rank 0's stdout; a failing rank makes the launcher exit non-zero.  (The ranks here are a stub script:
    no GPU in this container.)"""
    import subprocess
    import sys
    import bench
    stub = tmp_path / "stub.py"
    stub.write_text("import os,sys\n"
                    "r=int(os.environ['RANK']); w=int(os.environ['WORLD_SIZE'])\n"
                    "assert os.environ['MASTER_ADDR']=='127.0.0.1' and int(os.environ['MASTER_PORT'])>0\n"
                    "assert os.environ['LOCAL_RANK']==str(r)\n"
                    "if '--fail' in sys.argv and r==1: sys.exit(3)\n"
                    "print('{\"rank\": %d, \"world\": %d}' % (r, w))\n")
    code = ("import sys, bench, argparse; bench.__file__=%r; sys.argv=['bench.py']+sys.argv[1:];"
            "sys.exit(bench.launch(argparse.Namespace(gpus=2)))" % str(stub))
    env = dict(os.environ, PYTHONPATH=ROOT)
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stderr
    assert p.stdout.strip() == '{"rank": 0, "world": 2}'       # rank 0's line only; rank 1 goes to stderr
    assert '"rank": 1' in p.stderr
    p = subprocess.run([sys.executable, "-c", code, "--fail"], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode != 0


def test_native_callers_build_and_link():
    """The C99 caller and the C++ adaptor caller compile against include/ alone and link libdrfe.so (no compute here)."""
    import subprocess
    nat = os.path.join(ROOT, "tests", "native")
    subprocess.check_call(["make", "-C", nat], stdout=subprocess.DEVNULL)
    for exe in ("c_caller", "adaptor_caller"):
        out = subprocess.run(["ldd", os.path.join(nat, exe)], capture_output=True, text=True).stdout
        assert "libdrfe.so" in out and "not found" not in out.split("libdrfe.so")[1].split("\n")[0], out
    # without arguments both print nothing and exit 2 before touching the device
    assert subprocess.run([os.path.join(nat, "c_caller")]).returncode == 2


def test_lsd_host_stages_without_a_device(oracle_mod):
    """The sequential half of the product's LSD (pixel ordering, region growing, rectangle fit, deferred level-synchronous
    rect_improve with host pixel counts, NFA) on the oracle's level-line fields: same segments as the oracle's own
    sequential implementation.  No device involved (drfe_lsd_segments_host)."""
    from dr_slam_amd import lib, synth
    for seed, kind, mode in ((2, "room_boxes", 0), (5, "corridor", 0), (2, "room_boxes", 1), (5, "corridor", 1), (2, "room_boxes", 2), (1, "planar_lowtexture", 0)):
        g, _, _ = next(synth.sequence(seed, 1, kind=kind))
        o = oracle_mod.extract_lines(g, max_lines=100000, stages=True, rect_mode=mode)
        ang = o["angles"]
        cs = np.zeros(ang.shape + (2,), np.float32)
        defined = ang != -1024.0
        a32 = ang.astype(np.float32)
        # canonical cos/sin of float(angle): the exactly rounded value (oracle_math.h / drfe_math.h agree on it)
        cs[..., 0] = np.where(defined, np.cos(a32.astype(np.float64)).astype(np.float32), 0)
        cs[..., 1] = np.where(defined, np.sin(a32.astype(np.float64)).astype(np.float32), 0)
        segs = lib.lsd_segments_host(o["modgrad"], ang, cs, float(o["modgrad"].max()), rect_mode=mode)
        assert len(segs) == o["detected"] > 40
        h, w = g.shape
        # LSDDetector's checkLineExtremes: below 0 -> 0, >= size -> size - 1 (an end point inside (size - 1, size) stays)
        e = segs.copy()
        for col, lim in ((0, w), (2, w), (1, h), (3, h)):
            v = e[:, col]
            e[:, col] = np.where(v < 0, np.float32(0), np.where(v >= np.float32(lim), np.float32(lim) - np.float32(1), v))
        L = o["lines"]
        ref = np.stack([L["startPointX"], L["startPointY"], L["endPointX"], L["endPointY"]], 1)
        assert np.array_equal(e.view(np.uint32), ref.view(np.uint32))


def test_ahc_host_stages_without_a_device(oracle_mod):
    """The host half of the product's AHC extractor (graph, clustering, membership, flood fill, re-merge, labels) on the oracle's
    block fits: same planes, label image and member lists as the oracle.  No device involved (drfe_planes_ahc_from_blocks)."""
    from dr_slam_amd import lib, synth
    for cam, kind, seed in ((synth.TUM3, "room_boxes", 2), (synth.ICL, "living_room", 3)):
        _, d, _ = next(synth.sequence(seed, 1, cam=cam, kind=kind))
        K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32)
        inv = float(np.float32(1.0) / np.float32(cam.depth_factor))
        o = oracle_mod.ahc_planes(d, K4, inv)
        g = lib.planes_ahc_from_blocks(o["blocks"], o["block_valid"], o["block_N"], d, K4, inv)
        assert len(g["planes"]) == len(o["planes"]) >= 2
        assert np.array_equal(g["seg"], o["seg"])
        assert np.array_equal(g["planes"]["normal"].view(np.uint64), o["planes"][:, 0:3].view(np.uint64))
        assert np.array_equal(g["planes"]["center"].view(np.uint64), o["planes"][:, 3:6].view(np.uint64))
        assert np.array_equal(g["planes"]["n_points"], o["N"])
        for a, b in zip(g["members"], o["members"]):
            assert np.array_equal(a, b)


def test_ahc_vector_trial_solver_equals_scalar():
    """ahc_math_simd.h: the 4-lane (AVX2) and 8-lane (AVX-512F) instantiations of the AHC trial-merge fit against the scalar
    routine, bit for bit, on point-cloud statistics of every kind the clustering meets and on degenerate ones."""
    import ctypes as C
    from dr_slam_amd import lib
    L = lib.load()
    rng = np.random.default_rng(5)
    recs, Ns = [], []

    def add(P):
        P = np.asarray(P, np.float64)
        x, y, z = P[:, 0], P[:, 1], P[:, 2]
        recs.append([x.sum(), y.sum(), z.sum(), (x * x).sum(), (y * y).sum(), (z * z).sum(), (x * y).sum(), (y * z).sum(), (x * z).sum()])
        Ns.append(len(P))

    for _ in range(3000):
        n = int(rng.integers(100, 30000))
        m = min(n, 400)
        nrm = rng.normal(size=3); nrm /= np.linalg.norm(nrm)
        a = np.cross(nrm, [1, 0, 0.3]); a /= np.linalg.norm(a); b = np.cross(nrm, a)
        ext = rng.uniform(0.05, 3.0, 2)
        P = rng.uniform(0.5, 4.0) * nrm + rng.uniform(-1, 1, (m, 1)) * ext[0] * a + rng.uniform(-1, 1, (m, 1)) * ext[1] * b \
            + rng.normal(0, rng.choice([0, 1e-4, 3e-3, 0.05]), (m, 1)) * nrm
        add(P)
        Ns[-1] = n                                   # sums of m points presented as N = n: exercises odd scalings too
    add(np.ones((100, 3)))                           # zero covariance
    add(np.zeros((100, 3)))                          # all-zero sums (scale == 0 branch)
    add(np.stack([np.linspace(0, 1, 100), np.zeros(100), np.ones(100)], 1))          # a line: two zero eigenvalues
    add(np.stack([np.linspace(0, 1, 100), np.linspace(0, 2, 100), np.full(100, 2.0)], 1))
    g = np.stack(np.meshgrid(np.arange(10.0), np.arange(10.0)), -1).reshape(-1, 2)
    add(np.concatenate([g, np.full((100, 1), 1.5)], 1))                             # exact axis-aligned plane (no Householder step)
    add(np.concatenate([np.full((100, 1), -2.0), g], 1))
    S = np.ascontiguousarray(recs, np.float64)
    N = np.ascontiguousarray(Ns, np.int32)
    out = {}
    for mode in (0, 1, 2):
        o = np.zeros((len(S), 8))
        rc = L.drfe_debug_ahc_trials(S.ctypes.data_as(C.c_void_p), N.ctypes.data_as(C.c_void_p), len(S), mode, o.ctypes.data_as(C.c_void_p))
        if rc == -4:
            continue                                  # this CPU lacks the vector width
        assert rc == 0
        out[mode] = o
    assert 0 in out and len(out) >= 2, "no vector instantiation available on this host"
    for mode, o in out.items():
        same = (o.view(np.uint64) == out[0].view(np.uint64)) | (np.isnan(o) & np.isnan(out[0]))
        assert same.all(), (mode, np.argwhere(~same)[:5])
    for n in (1, 2, 3, 5, 9):                         # ragged tails: fewer records than lanes
        o1, o0 = np.zeros((n, 8)), np.zeros((n, 8))
        for mode, o in ((max(k for k in out), o1), (0, o0)):
            assert L.drfe_debug_ahc_trials(S.ctypes.data_as(C.c_void_p), N.ctypes.data_as(C.c_void_p), n, mode, o.ctypes.data_as(C.c_void_p)) == 0
        assert np.array_equal(o1.view(np.uint64), o0.view(np.uint64))


def test_restated_introsort_equals_std_sort():
    """introsort_restated.h: the product's restatement of libstdc++'s introsort (mask-driven Hoare partitions + a stable final pass)
    against std::sort with the reference's comparator itself - the permutation of equal keys included - for both users (LSD's
    pseudo-ordering, VoxelGrid's index sort) on arrays of many sizes and key distributions; with a forced depth limit (heap-sort
    branch) against the plain transcription, which is itself compared with std::sort at the natural limit."""
    import ctypes as C
    from dr_slam_amd import lib
    L = lib.load()
    rng = np.random.default_rng(11)

    def run(recs, kind, mode, depth=-1, skip_below=0):
        k = recs.copy()
        rc = L.drfe_debug_order_sort(k.ctypes.data_as(C.c_void_p), len(k), kind, mode, depth, skip_below)
        if rc == -4:
            return None                               # no AVX2 on this CPU
        assert rc == 0
        return k

    def recs_of(keys, kind):
        keys = np.asarray(keys)
        if kind == 0:
            return ((keys.astype(np.uint32) << 22) | (np.arange(len(keys), dtype=np.uint32) & 0x3FFFFF)).astype(np.uint32)
        return ((keys.astype(np.uint64) << np.uint64(32)) | np.arange(len(keys), dtype=np.uint64)).astype(np.uint64)

    for kind, top in ((0, 1024), (1, 1 << 21)):
        cases = []
        for trial in range(420):
            n = int(rng.integers(0, 6000)) if trial % 3 else int(rng.integers(0, 300))
            shape = trial % 7
            if shape == 0: b = rng.integers(0, top, n)
            elif shape == 1: b = rng.integers(0, 4, n)
            elif shape == 2: b = np.full(n, 7)
            elif shape == 3: b = np.minimum(top - 1, rng.exponential(20, n).astype(int))
            elif shape == 4: b = np.sort(rng.integers(0, 50, n))
            elif shape == 5: b = np.sort(rng.integers(0, 50, n))[::-1]
            else: b = np.concatenate([np.arange(n // 2), np.arange(n - n // 2)[::-1]]) % top       # organ pipe
            cases.append(recs_of(b, kind))
        if kind == 0:
            cases.append(recs_of(np.minimum(1023, rng.exponential(30, 511 * 383).astype(int)), 0))    # one 512 x 384 field
        else:   # a plane's points in scan order: leaf indices locally repetitive, globally increasing with noise
            t = np.arange(60000)
            cases.append(recs_of((t // 640) // 12 * 200 + (t % 640) // 12 + rng.integers(0, 2, len(t)) * 40000, 1))
        for k in cases:
            want = run(k, kind, 0)
            for mode in (1, 2, 3):
                got = run(k, kind, mode)
                assert got is None or np.array_equal(got, want), (kind, len(k), mode)
            if kind == 0 and len(k):
                # only the bins >= skip_below wanted: that prefix is std::sort's, the rest holds the same keys bin by bin
                bins = k >> 22
                for skip in (int(np.median(bins)), int(bins.max()), 1):
                    keep = int((bins >= skip).sum())
                    for mode in (1, 2):
                        got = run(k, 0, mode, -1, skip)
                        if got is None:
                            continue
                        assert np.array_equal(got[:keep], want[:keep]), (len(k), mode, skip)
                        assert np.array_equal(got >> 22, want >> 22) and np.array_equal(np.sort(got), np.sort(want))
            if len(k) < 6000:
                for depth in (0, 1, 3):
                    want_d = run(k, kind, 3, depth)
                    for mode in (1, 2):
                        got = run(k, kind, mode, depth)
                        assert got is None or np.array_equal(got, want_d), (kind, len(k), mode, depth)


def test_cr_sincos_is_correctly_rounded():
    """dr_slam_amd/csrc/cr_sincos.h (the cos / sin of region2rect and of region_grow's seed direction on the device and host paths
    of the line detector): against a 60-digit decimal evaluation on 400 angles of the form the detector produces (float degrees
    x pi/180, optionally + pi) - the result must be the correctly rounded double - and against this host's libm on 200 000
    (glibc stays within 0.55 ulp: it may differ by one ulp, rarely, and never in the float the seed direction is rounded to)."""
    import ctypes as C, math
    from decimal import Decimal, getcontext
    from fractions import Fraction
    from dr_slam_amd import lib
    L = lib.load()
    rng = np.random.default_rng(7)
    deg = rng.uniform(0, 360, 200000).astype(np.float32)
    x = deg.astype(np.float64) * (math.pi / 180.0)
    x[1::2] += math.pi
    x[:6] = [0.0, math.pi / 2, math.pi, 1.5 * math.pi, 2 * math.pi, 3 * math.pi]
    s = np.zeros_like(x); c = np.zeros_like(x); ok = np.zeros(len(x), np.int32)
    assert L.drfe_debug_cr_sincos(x.ctypes.data_as(C.c_void_p), len(x), s.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p),
                                  ok.ctypes.data_as(C.c_void_p)) == 0
    assert ok.all()
    gs = np.array([math.sin(v) for v in x]); gc = np.array([math.cos(v) for v in x])
    ulp_s = np.abs(s - gs) / np.spacing(np.abs(gs) + 1e-300); ulp_c = np.abs(c - gc) / np.spacing(np.abs(gc) + 1e-300)
    assert ulp_s.max() <= 1 and ulp_c.max() <= 1
    assert (ulp_s > 0).mean() < 1e-2 and (ulp_c > 0).mean() < 1e-2
    assert np.array_equal(s.astype(np.float32), gs.astype(np.float32)) and np.array_equal(c.astype(np.float32), gc.astype(np.float32))
    getcontext().prec = 70

    def dec(v):
        f = Fraction(float(v))
        return Decimal(f.numerator) / Decimal(f.denominator)

    def series(X, first, k0):
        term, tot, n = first, first, k0
        while abs(term) > Decimal(10) ** -66:
            term = -term * X * X / (n * (n + 1)); tot += term; n += 2
        return tot
    # every libm disagreement plus a random sample: the routine's value must be the double nearest to the exact one
    idx = np.unique(np.concatenate([np.nonzero((ulp_s > 0) | (ulp_c > 0))[0][:100], rng.integers(0, len(x), 300)]))
    for i in idx:
        X = dec(x[i])
        for exact, got in ((series(X, X, 2), s[i]), (series(X, Decimal(1), 1), c[i])):
            err = abs(dec(got) - exact)
            for nb in (np.nextafter(got, np.inf), np.nextafter(got, -np.inf)):
                assert err <= abs(dec(nb) - exact)


def test_shard_sequences_dealt_round_robin():
    """drfe_shard_sequences_of_rank (host arithmetic of the native multi-GPU helper): every sequence goes to exactly one rank."""
    from dr_slam_amd import lib
    for nseq, nranks in ((8, 8), (8, 3), (5, 8), (0, 2), (17, 4)):
        got = [lib.Shard.sequences_of_rank(nseq, nranks, r) for r in range(nranks)]
        assert sorted(int(v) for g in got for v in g) == list(range(nseq))
        assert all((g % nranks == r).all() for r, g in enumerate(got))
    with pytest.raises(lib.DrfeError):
        lib.Shard.sequences_of_rank(4, 2, 2)


def test_long_kernels_keep_their_lds_budget(tmp_path):
    """The full front-end is bound by LDS x time: how many of the long kernels' workgroups a CU holds is
    decided by their LDS against the 1280-byte allocation granule of its 160 KB.  Reads the kernels' static LDS out of the code
    objects inside libdrfe.so (llvm-objdump --offloading + llvm-readelf --notes; no GPU) and holds each to the residency the
    design counts on; the dynamic LDS of the two sorts is introsort_device.h's ORD_DYN_WORDS x threads x 4."""
    import re, shutil, subprocess
    llvm = "/opt/rocm/lib/llvm/bin"
    so = os.path.join(ROOT, "dr_slam_amd", "csrc", "libdrfe.so")
    if not (os.path.exists(so) and os.path.exists(os.path.join(llvm, "llvm-objdump")) and os.path.exists(os.path.join(llvm, "llvm-readelf"))):
        pytest.skip("libdrfe.so or the LLVM tools are not here")
    work = tmp_path / "co"
    work.mkdir()
    shutil.copy(so, work / "x.so")
    subprocess.run([os.path.join(llvm, "llvm-objdump"), "--offloading", "x.so"], cwd=work, check=True, capture_output=True)
    lds = {}
    for f in sorted(os.listdir(work)):
        if "gfx950" not in f:
            continue
        notes = subprocess.run([os.path.join(llvm, "llvm-readelf"), "--notes", f], cwd=work, capture_output=True, text=True).stdout
        size = None
        for line in notes.splitlines():
            m = re.match(r"\s+\.group_segment_fixed_size:\s+(\d+)", line)
            if m:
                size = int(m.group(1))
            m = re.match(r"\s+\.name:\s+(\S+)", line)
            if m and size is not None:
                lds[m.group(1)] = size
                size = None
    def find(sub, exact=False):
        hits = [v for k, v in lds.items() if (k == sub if exact else sub in k)]
        assert len(hits) == 1, (sub, sorted(lds))
        return hits[0]
    granule, cu = 1280, 160 * 1024
    def per_cu(nbytes):
        return cu // (-(-nbytes // granule) * granule)
    words = int(re.search(r"#define ORD_DYN_WORDS (\d+)", open(os.path.join(ROOT, "dr_slam_amd", "csrc", "introsort_device.h")).read()).group(1))
    assert per_cu(find("k_ahc_cluster", True)) >= 7
    assert per_cu(find("k_ahc_refine", True)) >= 9
    # the 1280 x 960 instantiations (12 800-entry queue, 1024-entry lists): one / five frames per CU, the queue still all in LDS
    assert per_cu(find("k_ahc_cluster_big", True)) >= 1 and per_cu(find("k_ahc_refine_big", True)) >= 4
    assert per_cu(find("k_rect_improve")) >= 7
    assert per_cu(find("k_lsd_order") + words * 256 * 4) >= 7
    assert per_cu(find("k_voxel_grid") + words * 256 * 4) >= 7
    # k_lsd_grow: all dynamic - bitmap of the 0.8-scaled 640 x 480 frame + member ring + the three-sum columns (drfe_lsd_grow_lds_bytes)
    ring = int(re.search(r"#define LSD_RING (\d+)", open(os.path.join(ROOT, "dr_slam_amd", "csrc", "lsd_grow_kernels.hip")).read()).group(1))
    assert find("k_lsd_grow", True) == 0 and per_cu(512 * 384 // 8 + ring * 4 + 64 * 3 * 8) >= 6
    assert find("k_lsd_grow_mw", True) == 0            # dynamic LDS: bitmap + control block + 16 overlay tables + 4 x (ring + columns) = 46.8 KB at 512 x 384: three workgroups per CU


def test_bench_refuses_the_sharded_full_frontend_without_three_cpus_per_rank():
    """`bench.py --gpus N --full-frontend` keeps ~2.5 host cores busy per GPU (upload threads, pools, Python): with fewer than 3 CPUs per
    rank it would report a number bound by the host, so the launcher refuses (exit 3) before any rank - or any GPU call - starts."""
    import subprocess
    import sys
    from dr_slam_amd import sharding
    n = sharding.host_cpus() // 3 + 1
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--full-frontend", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 3, (p.returncode, p.stderr[-400:])
    assert "3 per rank" in p.stderr and p.stdout.strip() == ""
