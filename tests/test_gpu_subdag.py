"""Several GPUs behind ONE public_ctx.execute() (eva_amd/host/multi_device.h): the mode is
chosen inside execute(), as the reference chooses its parallel traversal inside SEALPublic::execute
(/root/reference/eva/seal/seal.cpp:105-113).

  sub-DAG split   Harris' three convolutions (/root/reference/examples/image_processing.py:65-100) as
                  three evah_execute submits on three members, evah_ct_copy at the cuts
  limb sharding   RNS limbs dealt over the members; all-gather + broadcast per key switch
  dag             execute_batch deals the groups of a batch over the members

Members are device indices; a repeated index gives several contexts on the one GPU of this box (the
cross-device branch runs when a second device is visible, see test_two_real_devices).  Every result must
equal the oracle's walk of the same DAG and the single-device execute(), bit for bit."""
import numpy as np
import pytest

from eva import EvaProgram, Input, Output
from eva.ckks import CKKSCompiler
from eva.seal import generate_keys
from eva_amd import backend
from oracle_executor import c_walk
from eva_amd.workloads import harris as _harris, image as _image

pytestmark = pytest.mark.gpu


def _same_as(out, ref, single=None):
    for name in ref:
        got = out.get(name)
        assert np.array_equal(got[4], ref[name]), f"{name}: differs from the oracle walk"
        if single is not None:
            assert np.array_equal(got[4], single.get(name)[4]) and got[3] == single.get(name)[3]


@pytest.mark.parametrize("members", [[0, 0], [0, 0, 0]], ids=["2", "3"])
def test_harris_subdag_split_through_execute(members):
    compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(_harris())
    params.poly_modulus_degree = 16384
    pub, sec = generate_keys(params, 4)
    enc = pub.encrypt(_image(4096), sig)
    single = pub.execute(compiled, enc)
    ref, _ = c_walk(pub, compiled, enc, threads=4)
    pub.devices, pub.shard_mode = members, "subdag"
    out = pub.execute(compiled, enc)
    plan = pub.last_subdag_plan
    comps = plan[1:-1]
    assert len(comps) >= 2 and sorted({m for m, _ in comps}) == list(range(len(members)))[:len({m for m, _ in comps})]
    if len(members) == 3:  # the three convolutions: equal pieces, one per member
        assert sorted(m for m, _ in comps) == [0, 1, 2] and len({n for _, n in comps}) == 1
    _same_as(out, ref, single)
    again = pub.execute(compiled, enc)  # the device group is reused
    _same_as(again, ref)
    res = sec.decrypt(again, sig)       # resident outputs live on member 0's device state
    assert np.isfinite(np.array(res['image'])).all()
    pub.devices, pub.shard_mode = [], ""
    _same_as(pub.execute(compiled, enc), ref)


def test_program_without_parallel_branches_stays_on_one_member():
    prog = EvaProgram('chain', vec_size=64)
    with prog:
        x = Input('x')
        Output('y', (x * x + x) * x)
    prog.set_input_scales(30)
    prog.set_output_ranges(20)
    compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(prog)
    pub, sec = generate_keys(params, 2, devices=[0, 0], shard="subdag")
    enc = pub.encrypt({'x': [i / 64.0 for i in range(64)]}, sig)
    out = pub.execute(compiled, enc)
    assert [n for _, n in pub.last_subdag_plan[1:-1]] == []  # nothing worth cutting: one piece
    ref, _ = c_walk(pub, compiled, enc)
    _same_as(out, ref)


def _conv_chain(depth):
    prog = EvaProgram('conv+chain', vec_size=1024)
    with prog:
        x = Input('x')
        acc = None
        for i in range(3):
            t = (x << i) * ([0.25 * (i + 1)] * 1024)
            acc = t if acc is None else acc + t
        for _ in range(depth):
            acc = acc * acc
        Output('y', acc)
    prog.set_input_scales(30)
    prog.set_output_ranges(20)
    return CKKSCompiler(config={'warn_vec_size': 'false'}).compile(prog)


@pytest.mark.parametrize("G", [2, 3, 4])
def test_limb_sharded_execute_bit_exact(G):
    compiled, params, sig = _conv_chain(3)
    params.poly_modulus_degree = 8192
    pub, sec = generate_keys(params, 6)
    rng = np.random.default_rng(G)
    inputs = {'x': list(rng.uniform(-1, 1, 1024))}
    enc = pub.encrypt(inputs, sig)
    single = pub.execute(compiled, enc)
    ref, _ = c_walk(pub, compiled, enc)
    pub.devices, pub.shard_mode = [0] * G, "limb"
    for call in range(2):  # the second call finds the program's plaintexts already dealt over the shards
        out = pub.execute(compiled, enc)
        _same_as(out, ref, single)
    assert pub.last_exchanged_words > 0
    # one launch per receiving shard per exchange step: a key switch = all-gather (G) + broadcast (G - 1), a rescale
    # = broadcast (G - 1) — not the G (G - 1) + (G - 1) copies of the r03 exchange
    kinds = [(str(d["op"]).split(".")[-1], d) for d in compiled._dump()]
    n_ks = sum(1 for kk, d in kinds if kk == "Relinearize" or (kk in ("RotateLeftConst", "RotateRightConst") and d.get("rotation", 1) != 0))
    n_rs = sum(1 for kk, _ in kinds if kk == "Rescale")
    assert n_ks > 0 and n_rs > 0
    assert pub.last_exchange_launches == n_ks * (2 * G - 1) + n_rs * (G - 1), (pub.last_exchange_launches, n_ks, n_rs, G)
    # every shard holds its own prime rows of the keys and nothing else: its data limbs + the special prime
    kb = pub.key_bytes()
    k = len(list(pub.primes))
    assert len(kb) == G + 1 and kb[-1] > 0
    for s in range(G):
        rows = len(range(s, k - 1, G)) + 1
        assert kb[s] * k == kb[-1] * rows, (s, kb, rows, k)
    assert sum(kb[:G]) == kb[-1] // k * (k - 1 + G)
    res = sec.decrypt(out, sig)
    from eva import evaluate
    want = evaluate(compiled, inputs)
    assert np.abs(np.array(res['y']) - np.array(want['y'])).max() < 1e-2


def test_limb_sharded_context_never_uploads_the_whole_keys():
    """limb mode chosen at key generation: the whole evaluation keys are never resident on one device"""
    compiled, params, sig = _conv_chain(2)
    params.poly_modulus_degree = 8192
    pub, sec = generate_keys(params, 6, devices=[0, 0], shard="limb")
    enc = pub.encrypt({'x': [0.5] * 1024}, sig)
    out = pub.execute(compiled, enc)
    kb = pub.key_bytes()
    assert kb[-1] == 0 and all(b > 0 for b in kb[:2])
    ref, _ = c_walk(pub, compiled, enc)
    for name in out.names():
        assert np.array_equal(out.get(name)[4], ref[name])
    # library rules: a shard's rows cannot serve the unsharded entry points, and its map is fixed once keys are up
    primes = [int(q) for q in pub.primes]
    k, N = len(primes), 8192
    ctx = backend.Context(N, primes)
    ctx.set_shard(1, 2)
    ctx.upload_relin_key(np.zeros((k - 1, 2, k, N), dtype=np.uint64))
    assert ctx.key_bytes() == (k - 1) * 2 * (len(range(1, k - 1, 2)) + 1) * N * 8
    ctx.set_shard(1, 2)
    with pytest.raises(RuntimeError, match="current shard map"):
        ctx.set_shard(0, 2)
    ct = ctx.upload_ct(np.zeros((3, 1, N), dtype=np.uint64), 2.0 ** 30)
    with pytest.raises(RuntimeError, match="limb shard's key rows"):
        ctx.relinearize(ct)


def test_dag_mode_deals_a_batch_over_the_members():
    from eva_amd.workloads import sobel as _sobel
    prog = _sobel(64, 64, 4096)
    prog.set_input_scales(25)
    prog.set_output_ranges(10)
    compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(prog)
    params.poly_modulus_degree = 8192
    pub, sec = generate_keys(params, 3)
    pub.resident = False
    encs = [pub.encrypt({'image': [((37 * i + u) % 256) / 255.0 for i in range(4096)]}, sig) for u in range(5)]
    batch = [encs[i % 5] for i in range(23)]
    pub.batch_chunk = 4
    one = pub.execute_batch(compiled, batch)
    pub.devices, pub.shard_mode = [0, 0, 0], "dag"
    many = pub.execute_batch(compiled, batch)
    for a, b in zip(one, many):
        assert np.array_equal(a.get('image')[4], b.get('image')[4])
    ref, _ = c_walk(pub, compiled, batch[7])
    assert np.array_equal(many[7].get('image')[4], ref['image'])


@pytest.mark.parametrize("depth", [0, 2, 3, 8])
def test_groups_in_flight_do_not_change_a_batch(depth):
    """execute_batch rotates its groups over `batch_depth` issue queues (copies of one group against kernels of the
    others): any depth gives the ciphertexts of the one-by-one execute(), which equal the oracle walk"""
    from eva_amd.workloads import sobel as _sobel
    prog = _sobel(64, 64, 4096)
    prog.set_input_scales(25)
    prog.set_output_ranges(10)
    compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(prog)
    params.poly_modulus_degree = 8192
    pub, sec = generate_keys(params, 3)
    pub.resident = False
    encs = [pub.encrypt({'image': [((41 * i + 7 * u) % 256) / 255.0 for i in range(4096)]}, sig) for u in range(4)]
    batch = [encs[(3 * i) % 4] for i in range(19)]
    pub.batch_chunk = 2          # ten groups: every queue is reused, the last group is ragged
    pub.batch_depth = depth
    outs = pub.execute_batch(compiled, batch)
    assert len(outs) == len(batch)
    singles = [pub.execute(compiled, e) for e in encs]
    for i, o in enumerate(outs):
        assert np.array_equal(o.get('image')[4], singles[(3 * i) % 4].get('image')[4]), i
    ref, _ = c_walk(pub, compiled, batch[5])
    assert np.array_equal(outs[5].get('image')[4], ref['image'])
    # group sizes: full groups and a remainder (5 5 5 4 for 19 in groups of at most 5) or balanced (batch_balance, r6) — the same words
    pub.batch_chunk = 5
    for balance in (False, True):
        pub.batch_balance = balance
        again = pub.execute_batch(compiled, batch)
        for i, o in enumerate(again):
            assert np.array_equal(o.get('image')[4], outs[i].get('image')[4]), (balance, i)
    pub.batch_depth = 1
    with pytest.raises(RuntimeError, match="batch_depth"):
        pub.execute_batch(compiled, batch)


@pytest.mark.skipif(backend.device_count() < 2, reason="needs two HIP devices: the cross-device (xGMI peer copy) branch of evah_ct_copy")
def test_two_real_devices():
    """sub-DAG split and limb sharding with members on DIFFERENT GPUs: peer copies at the cuts / in the exchanges"""
    compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(_harris())
    params.poly_modulus_degree = 16384
    pub, sec = generate_keys(params, 4)
    enc = pub.encrypt(_image(4096), sig)
    ref, _ = c_walk(pub, compiled, enc, threads=4)
    pub.devices, pub.shard_mode = [0, 1], "subdag"
    _same_as(pub.execute(compiled, enc), ref)
    assert {m for m, _ in pub.last_subdag_plan[1:-1]} == {0, 1}
    pub.devices, pub.shard_mode = [0, 1], "limb"
    _same_as(pub.execute(compiled, enc), ref)
    # the raw C-ABI copy between two device states
    g0 = backend.Context(pub.poly_modulus_degree, list(pub.primes), device=0)
    g1 = backend.Context(pub.poly_modulus_degree, list(pub.primes), device=1)
    data = enc.get('image')[4]
    a = g0.upload_ct(data, 2.0 ** 30)
    assert np.array_equal(g1.copy_here(a).download(), data)
    g1.close(); g0.close()


def test_members_with_device_states_of_their_own(monkeypatch):
    """Member ids d + 256 v are further device STATES on device d (eva_amd/host/multi_device.h): own tables, own keys, own
    queues.  [0, 256, 512] on the one GPU of this box takes what three GPUs take — a root per member with the keys
    uploaded to each, evah_ct_copy / evah_pt_copy between the queues of different states at the cuts, the cross-state
    ordering (events recorded under the signaller's device, waited for under the waiter's) — everything except the
    peer hardware (test_two_real_devices, hardware-only)."""
    compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(_harris())
    params.poly_modulus_degree = 16384
    pub, sec = generate_keys(params, 4)
    enc = pub.encrypt(_image(4096), sig)
    single = pub.execute(compiled, enc)
    ref, _ = c_walk(pub, compiled, enc, threads=4)
    pub.devices, pub.shard_mode = [0, 256, 512], "subdag"
    for _ in range(2):
        out = pub.execute(compiled, enc)
        _same_as(out, ref, single)
    assert sorted(m for m, _ in pub.last_subdag_plan[1:-1]) == [0, 1, 2]
    res = sec.decrypt(out, sig)  # the outputs came back to member 0's state, where the secret half reads them
    assert np.isfinite(np.array(res['image'])).all()
    # limb shards on separate states of one device (each shard has a state of its own in any case) and of "two"
    pub.devices, pub.shard_mode = [0, 256], "limb"
    _same_as(pub.execute(compiled, enc), ref)
    # dag mode: the groups of a batch dealt over two states
    pub2, _ = generate_keys(params, 4, devices=[0, 256], shard="dag")
    pub2.resident = False
    pub2.batch_chunk = 2
    encs = [pub2.encrypt(_image(4096), sig) for _ in range(5)]
    outs = pub2.execute_batch(compiled, encs)
    for e, o in zip(encs, outs):
        r, _ = c_walk(pub2, compiled, e, threads=4)
        _same_as(o, r)
    # the raw C-ABI copy between two device states, both ways, and the release of a value through the other state's queue
    g0 = backend.Context(pub.poly_modulus_degree, list(pub.primes), device=0)
    g1 = backend.Context(pub.poly_modulus_degree, list(pub.primes), device=0)
    data = enc.get('image')[4]
    a = g0.upload_ct(data, 2.0 ** 30)
    b = g1.copy_here(a)
    c2 = g0.copy_here(g1.add(b, b))
    assert np.array_equal(b.download(), data)
    assert np.array_equal(c2.download(), g0.add(a, a).download())
    g1.close(); g0.close()


def test_peer_access_refusal_is_an_error(monkeypatch):
    """evah_ctx_enable_peer: a pair the runtime refuses (hipDeviceCanAccessPeer says no) is an error naming the pair, never a
    silent staging through host memory.  On one GPU the refusal is provoked by asking about the device itself
    (EVAH_PEER_SELF_CHECK=1); the granting path (hipDeviceEnablePeerAccess) needs two GPUs."""
    monkeypatch.setenv("EVAH_PEER_SELF_CHECK", "1")
    compiled, params, sig = _conv_chain(1)
    params.poly_modulus_degree = 4096
    pub, sec = generate_keys(params, 6, devices=[0, 256], shard="subdag")
    enc = pub.encrypt({'x': [0.5] * 1024}, sig)
    with pytest.raises(RuntimeError, match="cannot access device 0 as a peer"):
        pub.execute(compiled, enc)
    g0 = backend.Context(4096, list(pub.primes), device=0)
    g1 = backend.Context(4096, list(pub.primes), device=0)
    with pytest.raises(backend.EvaHipError, match="as a peer"):
        g0.enable_peer(g1)
    g1.close(); g0.close()
