#!/usr/bin/env python3
"""Writes the byte-comparison reference of bench.py's bootstrap leg for runs with several ranks: the sha256 of the stock backend's
(oracle/_ref: the reference's own DCRTPoly) bootstrap of ciphertext 0 of a `total`-ciphertext batch at BASELINE configs[3]'s shape —
the same program, PRNG and OpenMP team as bench.py's rank 0 (openfhe_amd/boot_batch.py run_rank; all `total` ciphertexts are encrypted
so that the random streams agree, ciphertext 0 is bootstrapped and dumped).  With one rank bench.py makes this comparison live; with 2,
4 or 8 ranks the other ranks would wait minutes for rank 0's host run, so the digest is committed instead
(tests/golden/stock_bootstrap_digests.json; FHE_BENCH_STOCK_AT_SCALE=1 still runs it live).

  python tools/stock_boot_digest.py [total[:first] ...]   default: 64:3 128 256 512 (1 rank: the first 3 ciphertexts, as bench.py compares
                                                         them live — the committed entry cross-checks hosts; 2, 4, 8 ranks x 64: ciphertext 0)
"""
import hashlib
import json
import os
import platform
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from openfhe_amd import boot_batch as bb
    stock = os.path.join(ROOT, "tests", "hal", "_build", "libfhe_boot_batch_stock.so")
    prng = os.path.join(ROOT, "tests", "hal", "_build", "libdetprng.so")
    jobs = [(int(v.split(":")[0]), int(v.split(":")[1]) if ":" in v else 1) for v in (sys.argv[1:] or ["64:3", "128", "256", "512"])]
    logN, slots, key_threads = 17, 1 << 16, 8
    path = os.path.join(ROOT, "tests", "golden", "stock_bootstrap_digests.json")
    table = json.load(open(path)) if os.path.exists(path) else {}
    for total, first in jobs:
        t0 = time.time()
        dump = os.path.join(tempfile.mkdtemp(prefix="fhe_stockdigest_"), "ct0.bin")
        r = bb.run_rank(logN, slots, total, 1, 1, 0, prng, so=stock, dump_path=dump, warmup=0, key_threads=key_threads, keep=first,
                        eval_threads=os.cpu_count() or 1)
        r.pop("handle").close()
        digest = hashlib.sha256(open(dump, "rb").read()).hexdigest()
        key = f"logN{logN}_slots{slots}_total{total}_team{key_threads}_first{first}"
        table[key] = {"sha256": digest, "bytes": os.path.getsize(dump),
                      "made_by": f"tools/stock_boot_digest.py on {platform.node() or 'the build container'} ({os.cpu_count()} cores), "
                                 f"{time.time() - t0:.0f} s, stock backend oracle/_ref"}
        os.remove(dump)
        json.dump(table, open(path, "w"), indent=1, sort_keys=True)
        print(key, digest, f"{time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    main()
