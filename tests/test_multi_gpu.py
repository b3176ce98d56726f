"""The multi-device index of the C ABI (vaqhip_multi_*) on ONE GPU: logical shards on device 0
exercise the sharding, the exchange step (device copies, and RCCL itself with one rank) and the
merge; the result must equal the single index bit for bit.  Real multi-GPU runs are the
driver's; the per-rank form of the same path is tests/test_sharding_gloo.py."""
import numpy as np
import pytest

from helpers import assert_topk_matches, make_case

pytestmark = pytest.mark.gpu


def single(c):
    import vaq_amd
    v = vaq_amd.VaqHip()
    v.mBitsAlloc = list(c["bits"])
    v.mCentroidsPerSubs = c["cents"]
    v.mEigenVectors = c["eig"]
    v.mCodebook = c["codes"]
    return v


@pytest.mark.parametrize("bits,N", [([8] * 8, 300_000), ([12, 10, 9, 8, 8, 7, 6, 4], 100_000), ([8] * 16, 5)],
                         ids=["m8", "nonuniform", "fewer_rows_than_shards"])
@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0, 0, 0, 0, 0, 0]], ids=["1", "2", "8"])
def test_logical_shards_equal_single_index(vaqlib, oracle, bits, N, devices):
    from vaq_amd.index import VaqHipMulti
    c = make_case(4100 + len(devices), 128, bits, N, 19, dup_frac=0.02)
    k = 100
    ref = single(c).search(c["X"], k)
    m = VaqHipMulti(devices, c["bits"], c["cents"], c["eig"])
    m.set_codes(c["codes"])
    inf = m.info()
    per = (N + len(devices) - 1) // len(devices)
    assert inf["shard_rows"] == [max(0, min(N, (g + 1) * per) - min(N, g * per)) for g in range(len(devices))]
    for qb in (0, 1, 2):
        m.set_option("queries_per_pass", qb)
        a = m.search(c["X"], k)
        assert np.array_equal(a.labels, ref.labels) and np.array_equal(a.distances.view(np.uint32), ref.distances.view(np.uint32))
    inf = m.info()
    assert inf["exchange"] == (0 if len(devices) == 1 else 2)  # none / device copies (RCCL refuses duplicate GPUs)
    # against the oracle too (the single index is itself checked elsewhere)
    Xp = oracle.project(c["X"], c["eig"])
    o_lab, o_dis = oracle.search(Xp, c["cents"], c["codes"], k, max_bits=max(bits), projected=True, nthreads=8)
    ad = np.stack([oracle.all_dists(oracle.create_lut(Xp[q], c["cents"], max(bits)), c["codes"]) for q in range(19)])
    assert_topk_matches(a.labels.reshape(19, k), a.distances.reshape(19, k), o_lab, o_dis, ad, what="multi")
    m.close()


def test_rccl_binding_one_rank(vaqlib):
    """exchange = 1 runs ncclCommInitAll + ncclAllGather (one rank: a copy) on the GPU that is
    there: the RCCL binding of the C++ host is exercised end to end."""
    from vaq_amd.index import VaqHipMulti
    c = make_case(4200, 128, [8] * 8, 50_000, 7)
    ref = single(c).search(c["X"], 10)
    m = VaqHipMulti([0], c["bits"], c["cents"], c["eig"])
    m.set_codes(c["codes"])
    m.set_option("exchange", 1)
    a = m.search(c["X"], 10)
    assert m.info()["exchange"] == 1
    assert np.array_equal(a.labels, ref.labels) and np.array_equal(a.distances, ref.distances)
    import vaq_amd
    m2 = VaqHipMulti([0, 0], c["bits"], c["cents"], c["eig"])
    with pytest.raises(vaq_amd.VaqHipError):
        m2.set_option("exchange", 1)  # RCCL needs distinct devices
    m.close()
    m2.close()


def test_multi_append_ti_and_errors(vaqlib, oracle):
    from vaq_amd.index import VaqHipMulti
    import vaq_amd
    c = make_case(4300, 64, [8] * 8, 40_000, 9)
    m = VaqHipMulti([0, 0, 0], c["bits"], c["cents"], c["eig"])
    m.set_codes(c["codes"][:30_000])
    m.add_codes(c["codes"][30_000:])  # extends the last shard; labels continue
    assert m.info()["N"] == 40_000 and sum(m.info()["shard_rows"]) == 40_000
    ref = single(c).search(c["X"], 50)
    a = m.search(c["X"], 50)
    assert np.array_equal(a.labels, ref.labels) and np.array_equal(a.distances, ref.distances)
    # TI: every shard regroups its own rows under the same centres; visit = 1 and EA give the exact top-k
    rng = np.random.default_rng(5)
    pick = rng.integers(0, 40_000, size=20)
    cl = np.concatenate([c["cents"][s][c["codes"][pick, s].astype(np.int64)] for s in range(4)], axis=1)
    m.set_ti_clusters(cl, 4)
    m.set_method(vaq_amd.NNMethod.TI | vaq_amd.NNMethod.EA, 1.0)
    t = m.search(c["X"], 50)
    assert np.array_equal(np.sort(t.labels.reshape(9, 50), 1), np.sort(ref.labels.reshape(9, 50), 1))
    assert np.allclose(t.distances, np.sqrt(ref.distances), rtol=1e-6)
    with pytest.raises(vaq_amd.VaqHipError):
        VaqHipMulti([], c["bits"], c["cents"], c["eig"])
    m.close()


def test_a_failing_shard_does_not_hang_the_others(vaqlib):
    """One shard of four refuses the search (its method is set to TI without clusters): the call
    returns that shard's error, NO exchange step was enqueued for anybody (the collective is only
    issued after every shard succeeded: vaqhip_multi.cpp, multi_search_common), the index answers
    again once the shard is repaired, and destroy returns."""
    import ctypes as C
    from vaq_amd import _lib
    from vaq_amd.index import VaqHipMulti
    c = make_case(4300, 128, [8] * 8, 120_000, 11)
    ref = single(c).search(c["X"], 10)
    m = VaqHipMulti([0, 0, 0, 0], c["bits"], c["cents"], c["eig"])
    m.set_codes(c["codes"])
    L = _lib.load()
    L.vaqhip_index_set_method.argtypes = [C.c_void_p, C.c_uint, C.c_float]
    L.vaqhip_multi_shard.restype = C.c_void_p
    bad = C.c_void_p(m.shard(2))
    assert L.vaqhip_index_set_method(bad, 0x04 | 0x02, 1.0) == 0   # TI | EA, but no clusters were set
    for _ in range(3):
        with pytest.raises(_lib.VaqHipError) as e:
            m.search(c["X"], 10)
        assert "shard 2" in str(e.value)
    assert L.vaqhip_index_set_method(bad, 0x80, 1.0) == 0          # HEAP again
    a = m.search(c["X"], 10)
    assert np.array_equal(a.labels, ref.labels) and np.array_equal(a.distances.view(np.uint32), ref.distances.view(np.uint32))
    m.close()  # (joins the workers, synchronises and frees every shard: must return)


def test_multi_search_device_entry(vaqlib):
    """vaqhip_multi_search_device: queries and results stay on the device, the call only enqueues;
    several searches back to back on one stream, then one synchronisation."""
    import torch
    from vaq_amd.index import VaqHipMulti
    c = make_case(4400, 128, [8] * 16, 200_000, 33, dup_frac=0.02)
    ref = single(c)
    m = VaqHipMulti([0, 0, 0], c["bits"], c["cents"], c["eig"])
    m.set_codes(c["codes"])
    outs = []
    for n, k in ((33, 100), (5, 10), (33, 100), (17, 1)):
        q = torch.from_numpy(c["X"][:n]).cuda()
        outs.append((n, k, m.search_device(q, k)))
    torch.cuda.synchronize()
    for n, k, (l, d) in outs:
        r = ref.search(c["X"][:n], k)
        assert np.array_equal(l.cpu().numpy().ravel(), r.labels) and np.array_equal(d.cpu().numpy().ravel().view(np.uint32), r.distances.view(np.uint32))
    m.close()


@pytest.mark.parametrize("bits,N,nq", [([8] * 16, 1_200_000, 64), ([8] * 8, 900_000, 40)], ids=["m16", "m8"])
def test_staged_search_with_threshold_exchange(vaqlib, oracle, bits, N, nq):
    """vaqhip_search_begin_device / _finish_device on three row shards (three indexes on the one GPU):
    the thresholds each shard's first rounds leave are MIN-reduced (here with torch), every shard
    finishes under the global bound (less of it is in reach then; its list keeps what the first
    rounds found), and the merge of the three lists equals the single index bit for bit.  Also: no
    exchange (NULL), a second staged search on the same index, and the state errors."""
    import torch
    import vaq_amd
    from vaq_amd import _lib
    from vaq_amd.index import merge_topk_packed_device
    from vaq_amd.sharding import shard_bounds
    k, G = 50, 3
    c = make_case(4500 + len(bits), 8 * len(bits), bits, N, nq, dup_frac=0.02)
    ref = single(c)
    ref.set_option("bucket_major", 2)
    r = ref.search(c["X"], k)
    shards = []
    for g in range(G):
        lo, hi = shard_bounds(N, G, g)
        v = vaq_amd.VaqHip()
        v.mBitsAlloc = list(bits)
        v.mCentroidsPerSubs = c["cents"]
        v.mEigenVectors = c["eig"]
        v.mCodebook = c["codes"][lo:hi]
        v.id_base = lo
        v.set_option("bucket_major", 2)
        assert v.staged_supported(nq, k)
        shards.append(v)
    Xd = torch.from_numpy(c["X"]).cuda()
    for exchange in (True, False, True):
        packed = torch.empty((G, 2, nq, k), dtype=torch.int32, device="cuda")
        thr = torch.empty((G, nq), dtype=torch.int32, device="cuda")
        for g, v in enumerate(shards):
            v.search_begin_device(Xd, k, (packed[g, 0], packed[g, 1].view(torch.float32)), thr[g])
        # a second search while one is open is refused, and leaves the open one intact
        with pytest.raises(_lib.VaqHipError) as e:
            shards[0].search(c["X"], k)
        assert e.value.code == -7
        tmin = thr.min(dim=0).values.contiguous()
        assert (tmin >= 0).all()
        for g, v in enumerate(shards):
            v.search_finish_device(tmin if exchange else None)
        torch.cuda.synchronize()
        ml, md = merge_topk_packed_device(packed, G, nq, k)
        torch.cuda.synchronize()
        assert np.array_equal(ml.cpu().numpy().ravel(), r.labels)
        assert np.array_equal(md.cpu().numpy().ravel().view(np.uint32), r.distances.view(np.uint32))
    with pytest.raises(_lib.VaqHipError) as e:
        shards[0].search_finish_device(None)  # nothing open
    assert e.value.code == -7
    # a plan without the rounds cannot be staged
    small = single(dict(c, codes=c["codes"][:5000]))
    assert not small.staged_supported(nq, k)
    for v in shards:
        v.close()
