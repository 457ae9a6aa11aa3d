#!/usr/bin/env python3
"""TCP front-end mirroring the reference program (main.go:7-22: "Listens for a stream of H264 bytes on port 8000"):
accepts connections, decodes the Annex-B byte stream of each on the GPU and prints one line per batch of frames
(count + MD5 of the cropped I420 data); --out writes the raw frames.

    python tools/serve.py --port 8000 [--out frames.yuv]      # then e.g.:  nc 127.0.0.1 8000 < clip.h264
"""
import argparse
import hashlib
import os
import socket
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import h264decode_amd as H  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--host", default="0.0.0.0")
    ap.add_argument("--port", type=int, default=8000)
    ap.add_argument("--max-width", type=int, default=1920)
    ap.add_argument("--max-height", type=int, default=1088)
    ap.add_argument("--frames-per-batch", type=int, default=30)
    ap.add_argument("--display-order", type=int, default=0, metavar="DEPTH",
                    help="deliver frames in display order through a reorder buffer of this depth (streams with B pictures; single-connection mode)")
    ap.add_argument("--out", default=None, help="append decoded frames (tight I420) to this file")
    ap.add_argument("--once", action="store_true", help="serve one connection and exit")
    ap.add_argument("--batch", type=int, default=0, help="decode up to N concurrent connections side by side in one batched decoder (H.BatchServer)")
    args = ap.parse_args()
    srv = socket.socket()
    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    srv.bind((args.host, args.port))
    srv.listen(4)
    print("listening on %s:%d" % (args.host, args.port), flush=True)
    out = open(args.out, "ab") if args.out else None
    if args.batch > 0:  # several connections, one batched GPU decoder
        import select
        bs = H.BatchServer(max_connections=args.batch, max_width=args.max_width, max_height=args.max_height, frames_per_batch=args.frames_per_batch,
                           on_frames=lambda i, f: print("slot %d: %d frames md5 %s" % (i, len(f), hashlib.md5(f.tobytes()).hexdigest()), flush=True),
                           on_close=lambda i, n: print("slot %d closed after %d frames" % (i, n), flush=True))
        srv.setblocking(False)
        while True:
            if bs.active() < args.batch and select.select([srv], [], [], 0 if bs.active() else 0.5)[0]:
                conn, peer = srv.accept()
                print("%s -> slot %d" % (peer[0], bs.add(conn)), flush=True)
            if bs.active() and not bs.tick():
                select.select([c for c in bs.conn if c is not None], [], [], 0.01)
    while True:
        conn, peer = srv.accept()
        total = [0]

        def on_frames(frames):
            total[0] += len(frames)
            print("%s: %d frames (%d so far) md5 %s" % (peer[0], len(frames), total[0], hashlib.md5(frames.tobytes()).hexdigest()), flush=True)
            if out:
                out.write(frames.tobytes())

        try:
            H.ByteStreamReader(conn, on_frames=on_frames, max_width=args.max_width, max_height=args.max_height, frames_per_batch=args.frames_per_batch,
                               display_order=args.display_order)
        except H.H264MIError as e:
            print("%s: decode error: %s" % (peer[0], e), flush=True)
        if args.once:
            break


if __name__ == "__main__":
    main()
