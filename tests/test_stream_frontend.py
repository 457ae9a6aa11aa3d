"""Stream front-end: access-unit splitting of an Annex-B byte stream fed in arbitrary pieces, and the
connection handler that mirrors the reference's ByteStreamReader / handleConnection (h264/server.go:113-172)."""
import io
import random
import socket
import threading

import numpy as np
import pytest


def _streams(sg):
    a = sg.encode(width=64, height=48, frames=7, idr_period=3, profile_idc=77, cabac=1, slices=2, seed=5)[0]
    b = sg.encode(width=64, height=48, frames=4, idr_period=0, profile_idc=66, cabac=0, long_start_code=0, seed=6)[0]
    return a, b


def _count_units(H, chunk):
    """pictures in a chunk = slices with first_mb_in_slice == 0"""
    n = 0
    for nu in H.read_nal_units(chunk):
        if nu.Type in (1, 5) and nu.RBSP()[0] & 0x80:
            n += 1
    return n


@pytest.mark.parametrize("piece", [1, 7, 64, 1000, 1 << 20])
def test_splitter_is_independent_of_the_feeding_pattern(H, sg, piece):
    for stream in _streams(sg):
        sp = H.AccessUnitSplitter(max_units_per_chunk=3)
        chunks = []
        for i in range(0, len(stream), piece):
            chunks += sp.feed(stream[i:i + piece])
        chunks += sp.flush()
        assert b"".join(chunks) == stream  # nothing lost, nothing duplicated
        total = _count_units(H, stream)
        per_chunk = [_count_units(H, c) for c in chunks]
        assert sum(per_chunk) == total and max(per_chunk) <= 3 and min(per_chunk) >= 1
        for c in chunks:  # every chunk starts on a start code
            assert c[:3] == b"\x00\x00\x01" or c[:4] == b"\x00\x00\x00\x01"


def test_splitter_random_pieces_and_delimiters(H, sg):
    a, _ = _streams(sg)
    aud = b"\x00\x00\x00\x01\x09\xf0"
    sei = b"\x00\x00\x01\x06\x05\x01\xaa\x80"
    # re-assemble with an access unit delimiter and an SEI in front of every picture
    nals = H.read_nal_units(a)
    parts = []
    for nu in nals:
        raw = a[nu._c.offset - 0:nu._c.offset + nu._c.num_bytes]
        if nu.Type in (1, 5) and nu.RBSP()[0] & 0x80:
            parts += [aud, sei]
        parts.append(b"\x00\x00\x01" + raw)
    stream = b"".join(parts)
    rng = random.Random(1)
    sp = H.AccessUnitSplitter(max_units_per_chunk=1)
    chunks, i = [], 0
    while i < len(stream):
        k = rng.randint(1, 300)
        chunks += sp.feed(stream[i:i + k])
        i += k
    chunks += sp.flush()
    assert b"".join(chunks) == stream
    assert all(_count_units(H, c) == 1 for c in chunks)
    # every access unit opens with its delimiter, or with the parameter sets that precede the delimiter of an IDR picture
    assert all(c.startswith(aud) or (H.read_nal_units(c)[0].Type == 7) for c in chunks)


@pytest.mark.gpu
def test_gpu_connection_handler_over_a_socket(H, sg, oracle_mod):
    stream = sg.encode(width=176, height=144, frames=9, idr_period=4, profile_idc=77, cabac=1, slices=2, seed=21)[0]
    ref, info = oracle_mod.decode(stream, crop=True)
    srv = socket.socket()
    srv.bind(("127.0.0.1", 0))
    srv.listen(1)

    def sender():
        c = socket.create_connection(srv.getsockname())
        rng = random.Random(2)
        i = 0
        while i < len(stream):
            k = rng.randint(1, 4000)
            c.sendall(stream[i:i + k])
            i += k
        c.close()

    t = threading.Thread(target=sender)
    t.start()
    conn, _ = srv.accept()
    got = []
    n = H.ByteStreamReader(conn, on_frames=got.append, max_width=176, max_height=144, frames_per_batch=4)
    t.join()
    srv.close()
    assert n == 9
    assert np.array_equal(np.concatenate(got), ref)


@pytest.mark.gpu
def test_gpu_handle_connection_reads_file_objects(H, sg, oracle_mod):
    stream = sg.encode(width=64, height=48, frames=5, idr_period=0, profile_idc=66, cabac=0, long_start_code=0, seed=22)[0]
    ref, _ = oracle_mod.decode(stream, crop=True)
    got = []
    assert H.handleConnection(io.BytesIO(stream), on_frames=got.append, max_width=64, max_height=48, frames_per_batch=2, read_size=333) == 5
    assert np.array_equal(np.concatenate(got), ref)


@pytest.mark.gpu
def test_gpu_handle_connection_with_slice_groups_and_arbitrary_slice_order(H, sg):
    """The connection handler cuts access units with the decoder's own picture-boundary test: a Baseline stream with an explicit slice
    group map, two slices per group in shuffled order, and one whose box-out map moves from picture to picture, read 211 bytes at a time."""
    for kw in (dict(width=176, height=144, frames=7, idr_period=3, profile_idc=66, cabac=0, slice_groups=3, fmo_type=6, slices=2, aso=1, seed=61),
               dict(width=176, height=144, frames=6, idr_period=0, profile_idc=66, cabac=0, slice_groups=2, fmo_type=3, slices=2, aso=1, num_ref_frames=2, seed=62)):
        stream, rec, _ = sg.encode(**kw)
        got = []
        assert H.handleConnection(io.BytesIO(stream), on_frames=got.append, max_width=176, max_height=144, frames_per_batch=2, read_size=211) == kw["frames"]
        assert np.array_equal(np.concatenate(got), rec)


@pytest.mark.gpu
def test_gpu_batch_server_decodes_several_connections_side_by_side(H, sg, oracle_mod):
    cfgs = [dict(width=176, height=144, frames=7, idr_period=3, profile_idc=77, cabac=1, seed=31),
            dict(width=64, height=48, frames=5, idr_period=0, profile_idc=66, cabac=0, long_start_code=0, seed=32),
            dict(width=180, height=100, frames=4, idr_period=0, profile_idc=100, cabac=1, transform8x8=1, slices=2, seed=33)]
    streams = [sg.encode(**c)[0] for c in cfgs]
    refs = [oracle_mod.decode(s, crop=True)[0] for s in streams]
    got = {i: [] for i in range(3)}
    closed = {}
    srv = H.BatchServer(max_connections=4, max_width=192, max_height=144, frames_per_batch=3,
                        on_frames=lambda i, f: got[i].append(f), on_close=lambda i, n: closed.__setitem__(i, n))
    pairs = [socket.socketpair() for _ in streams]
    slots = [srv.add(b) for _, b in pairs]
    assert slots == [0, 1, 2]

    def sender(sock, data, seed):
        rng = random.Random(seed)
        i = 0
        while i < len(data):
            k = rng.randint(1, 3000)
            sock.sendall(data[i:i + k])
            i += k
        sock.close()

    threads = [threading.Thread(target=sender, args=(a, s, 40 + i)) for i, ((a, _), s) in enumerate(zip(pairs, streams))]
    for t in threads:
        t.start()
    counts = srv.run()
    for t in threads:
        t.join()
    assert counts[:3] == [7, 5, 4] and closed == {0: 7, 1: 5, 2: 4}
    for i in range(3):
        assert np.array_equal(np.concatenate(got[i]), refs[i]), i



def test_display_order_buffer(H, sg):
    """DisplayOrder turns the coding order of B streams (incl. pyramids and several IDR periods) back into display order."""
    from conftest import MATRIX
    for name in ("b_ibp_cabac", "b_ibbp_cavlc", "b_pyramid_cabac", "b_pyramid_implicit", "b_gop_intra_pcm", "cabac_IPP"):
        kw = MATRIX[name]
        sg.encode(**kw)
        pocs = [int(p) for p in sg.last_pocs()]
        buf = H.DisplayOrder(depth=4)
        shown, seq = [], -1
        for k, poc in enumerate(pocs):
            new_seq = poc == 0 and (k == 0 or kw.get("idr_period", 0) > 0)
            seq += new_seq
            shown += buf.push((seq, poc), poc, new_seq)
        shown += buf.flush()
        assert len(shown) == len(pocs) and shown == sorted(shown), (name, shown)


@pytest.mark.gpu
def test_gpu_reader_delivers_b_streams_in_display_order(H, sg):
    from conftest import MATRIX
    kw = MATRIX["b_gop_intra_pcm"]  # I B B B P ..., two IDR periods
    stream, rec, _ = sg.encode(**kw)
    pocs = [int(p) for p in sg.last_pocs()]
    got = []
    r = H.H264Reader(io.BytesIO(stream), on_frames=lambda f: got.append(f), max_width=kw["width"], max_height=kw["height"], frames_per_batch=5, display_order=4)
    assert r.run() == kw["frames"]
    got = np.concatenate(got)
    seq, cur = [], -1
    for p in pocs:
        cur += p == 0
        seq.append(cur)
    order = sorted(range(len(pocs)), key=lambda i: (seq[i], pocs[i]))
    W, Hc = (kw["width"] + 15) // 16 * 16, (kw["height"] + 15) // 16 * 16
    want = rec[order]
    if (W, Hc) != (kw["width"], kw["height"]):  # the reader delivers cropped frames
        pytest.skip("cropped geometry")
    assert np.array_equal(got, want)


def test_splitter_cuts_slice_group_and_aso_streams_at_picture_boundaries(H, sg):
    """With slice groups or arbitrary slice order the slice of macroblock 0 is not the first of its picture, and after memory management
    operation 5 two pictures can agree in frame_num and picture order count: the splitter applies the decoder's own test (parsed slice
    headers, h264mi_slice_starts_picture) and must find exactly the generator's access units, whatever the feeding pattern."""
    from conftest import MATRIX
    names = [n for n, kw in MATRIX.items() if kw.get("slice_groups") or kw.get("aso")] + ["slices3_cabac" if "slices3_cabac" in MATRIX else "cabac_IPP"]
    assert len(names) >= 12
    for name in names:
        stream, _, sizes = sg.encode(**MATRIX[name])
        for piece in (1 << 20, 97):
            sp = H.AccessUnitSplitter(max_units_per_chunk=1)
            chunks = []
            for i in range(0, len(stream), piece):
                chunks += sp.feed(stream[i:i + piece])
            chunks += sp.flush()
            assert b"".join(chunks) == stream
            assert [len(c) for c in chunks] == [int(x) for x in sizes], (name, piece)


@pytest.mark.gpu
def test_gpu_reader_and_server_decode_field_streams(H, sg):
    """Field pictures through the stream front-end (h264/server.go:113-166): access units are PICTURES there -- a field each --, batches cut wherever
    frames_per_batch says, so the two fields of a frame routinely fall into different batches; every frame must come out once and exact.  A connection
    that ends on a lone first field still delivers that frame (the reader ends its input with an end-of-stream NAL unit)."""
    from conftest import FIELD_MATRIX
    for name, per_batch in (("field_IP", 3), ("field_mixed_paff", 1), ("field_b_spatial", 4), ("field_bottom_first", 5)):
        kw = FIELD_MATRIX[name]
        stream, rec, _ = sg.encode(**kw)
        got = []
        r = H.H264Reader(io.BytesIO(stream), on_frames=lambda f: got.append(f), max_width=kw["width"], max_height=kw["height"], frames_per_batch=per_batch)
        assert r.run() == kw["frames"], name
        assert np.array_equal(np.concatenate(got), rec), name
    kw = dict(FIELD_MATRIX["field_IP"], slices=1)
    stream, rec, _ = sg.encode(**kw)
    cut = stream.rfind(b"\x00\x00\x01")
    cut -= 1 if stream[cut - 1] == 0 else 0
    got = []
    r = H.H264Reader(io.BytesIO(stream[:cut]), on_frames=lambda f: got.append(f), max_width=kw["width"], max_height=kw["height"], frames_per_batch=2)
    assert r.run() == kw["frames"]
    out = np.concatenate(got)
    W, Hc = kw["width"], kw["height"]
    assert np.array_equal(out[:-1], rec[:-1])
    y = out[-1][:W * Hc].reshape(Hc, W)
    assert np.array_equal(y[0::2], rec[-1][:W * Hc].reshape(Hc, W)[0::2]) and (y[1::2] == 128).all()
    # several connections at once, field-coded and progressive, one batch per tick
    names = ["field_IP", "cabac_IPP", "field_mixed_paff"]
    from conftest import FULL_MATRIX
    enc = [sg.encode(**FULL_MATRIX[n]) for n in names]
    frames = {}
    srv = H.BatchServer(max_connections=3, max_width=176, max_height=144, frames_per_batch=2, on_frames=lambda i, f: frames.setdefault(i, []).append(f))
    for e in enc:
        assert srv.add(io.BytesIO(e[0])) >= 0
    counts = srv.run()
    for i, n in enumerate(names):
        assert counts[i] == FULL_MATRIX[n]["frames"], n
        assert np.array_equal(np.concatenate(frames[i]), enc[i][1]), n
