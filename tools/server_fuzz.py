#!/usr/bin/env python3
"""BatchServer under random clients (not a test): several connections at a time, each sending a random recipe's stream in random
pieces -- some cut off mid-stream, some sending garbage, new clients taking over the slots of finished ones.  Every well-behaved client
must get exactly its generator's frames; the others must be closed with a status, not take anybody down.  Usage: server_fuzz.py [rounds]"""
import os, random, socket, sys, threading, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import streamgen
import h264decode_amd as H

sys.argv = sys.argv[:2]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "param_sweep.py")).read()
g = {"__file__": os.path.join(os.path.dirname(os.path.abspath(__file__)), "param_sweep.py")}
_argv = sys.argv
sys.argv = ["x", "0", "--seed", "97"]
exec(compile(src[:src.index("if GPU:\n    import h264decode_amd")], "sweep", "exec"), g)
sys.argv = _argv
draw = g["draw"]
rnd = random.Random(5)
bad_total = 0
t0 = time.time()
for r in range(N):
    got, closed = {}, {}
    srv = H.BatchServer(max_connections=4, max_width=208, max_height=160, frames_per_batch=rnd.randint(1, 5),
                        on_frames=lambda i, f: got.setdefault(i, []).append(f), on_close=lambda i, n: closed.__setitem__(i, n))
    clients = []
    for c in range(rnd.randint(3, 7)):
        while True:
            g["FIELDS"] = rnd.random() < 0.25  # a quarter of the clients send field pictures (PAFF): a frame is two access units there
            kw = draw()
            kw["width"], kw["height"] = 16 * rnd.randint(2, 13), 16 * rnd.randint(2, 10)  # (cropped output == coded size)
            if g["FIELDS"]:
                kw["height"] = 32 * rnd.randint(1, 5)
            else:
                kw.pop("interlace_sps", None)
            if max(1, kw.get("slices", 1)) * max(1, kw.get("slice_groups", 1)) > 16:
                continue
            try:
                s, rec, _ = streamgen.encode(**kw)
                break
            except RuntimeError:
                pass
        kind = rnd.choice(["good", "good", "good", "cut", "garbage"])
        clients.append((kw, s, rec, kind))
    pending = list(range(len(clients)))
    slot_of, threads, results = {}, [], {}

    def sender(sock, data, seed):
        q = random.Random(seed)
        i = 0
        try:
            while i < len(data):
                k = q.randint(1, 3000)
                sock.sendall(data[i:i + k])
                i += k
        except OSError:
            pass
        sock.close()

    def attach():
        while pending:
            a, b = socket.socketpair()
            slot = srv.add(b)
            if slot < 0:
                a.close(), b.close()
                return
            ci = pending.pop(0)
            kw, s, rec, kind = clients[ci]
            data = s if kind == "good" else (s[:rnd.randint(len(s) // 4, len(s) - 1)] if kind == "cut" else s[:200] + bytes(rnd.getrandbits(8) for _ in range(3000)))
            slot_of[ci] = (slot, len(got.get(slot, [])))
            t = threading.Thread(target=sender, args=(a, data, 100 + ci))
            t.start()
            threads.append(t)

    attach()
    frames_seen = {}
    while srv.active() or pending:
        if not srv.tick():
            time.sleep(0.0005)
        # harvest finished slots, hand them to waiting clients
        for ci, (slot, start) in list(slot_of.items()):
            if slot in closed and ci not in results:
                fr = got.get(slot, [])[start:]
                results[ci] = (np.concatenate(fr) if fr else np.zeros((0, 0), np.uint8), closed.pop(slot), srv.errors[slot])
        attach()
    for t in threads:
        t.join()
    for ci, (kw, s, rec, kind) in enumerate(clients):
        out, n, err = results[ci]
        if kind == "good":
            if not (n == kw["frames"] and out.shape == rec.shape and np.array_equal(out, rec)):
                bad_total += 1
                print("MISMATCH round %d client %d: n=%d err=%d %s" % (r, ci, n, err, kw), flush=True)
        else:
            k = min(len(out), len(rec))
            if kind == "cut" and k > 1 and not np.array_equal(out[:k - 1], rec[:k - 1]):  # all but possibly the last (cut) picture are exact
                bad_total += 1
                print("MISMATCH (cut) round %d client %d %s" % (r, ci, kw), flush=True)
    srv.decoder.close()
    print("round %d: %d clients, kinds %s, %.1fs" % (r + 1, len(clients), [c[3] for c in clients], time.time() - t0), flush=True)
print("server fuzz:", "OK" if not bad_total else "%d mismatches" % bad_total)
sys.exit(1 if bad_total else 0)
