"""Worker process of the directory drivers' decode pool (roomnet_amd/infer.py): `python -m roomnet_amd.decode_worker`.

Reads one JSON-encoded file path per line on stdin, decodes it (roomnet_amd.imageio.imread: the stand-in for the reference's
cv2.imread, infer.py:81) into a shared-memory block and answers one line on stdout: `name h w c`, or `-` for an unreadable
file.  Started with subprocess (a fresh interpreter: nothing of the parent's GPU state is forked, and -- unlike
multiprocessing's spawn -- the parent's __main__ module is not imported again, so a driver script without a
`if __name__ == "__main__"` guard stays safe)."""
import json
import sys


def main():
    from roomnet_amd.imageio import decode_to_shm
    out = sys.stdout
    for line in sys.stdin:
        line = line.strip()
        if not line:
            continue
        try:
            got = decode_to_shm(json.loads(line))
        except Exception:
            got = None
        out.write("-\n" if got is None else "%s %d %d %d\n" % (got[0], got[1][0], got[1][1], got[1][2]))
        out.flush()


if __name__ == "__main__":
    main()
