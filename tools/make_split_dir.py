"""Write a seeded synthetic knowledge graph as a transductive split directory in the reference's layout (train.txt / valid.txt /
test.txt of ``h<TAB>r<TAB>t`` lines, /root/reference/ultra/dataset.py:33-96) -- to exercise ``bench.py --data DIR [--ckpt PATH]``
and ``data.task_from_split_dir`` where no real dataset exists (the build machines have no network).

    python tools/make_split_dir.py OUT_DIR [--shape S-codexs] [--valid 1000] [--test 1000] [--ckpt]
--ckpt also writes OUT_DIR/td_ultra_like.pth: seeded random-init weights in the reference's checkpoint layout."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--shape", default="S-codexs")
    ap.add_argument("--valid", type=int, default=1000)
    ap.add_argument("--test", type=int, default=1000)
    ap.add_argument("--ckpt", action="store_true")
    args = ap.parse_args()
    from ultra_torchdrug_amd.checkpoint import save_checkpoint
    from ultra_torchdrug_amd.data import DEFAULT_SEED, SHAPES, synthetic_triples
    from ultra_torchdrug_amd.task import build_ultra
    n, n_fact, r = SHAPES[args.shape]
    triples, _, _ = synthetic_triples((n, n_fact + args.valid + args.test, r), DEFAULT_SEED)
    os.makedirs(args.out, exist_ok=True)
    bounds = [0, n_fact, n_fact + args.valid, len(triples)]
    for i, name in enumerate(("train.txt", "valid.txt", "test.txt")):
        with open(os.path.join(args.out, name), "w") as f:
            for h, t, rel in triples[bounds[i]:bounds[i + 1]]:
                f.write("/m/e%d\t/rel/r%d\t/m/e%d\n" % (h, rel, t))
    if args.ckpt:
        torch.manual_seed(DEFAULT_SEED)
        save_checkpoint(build_ultra(r), os.path.join(args.out, "td_ultra_like.pth"))
    print("wrote %s: %d / %d / %d triples, %d entities, %d relations" % (args.out, n_fact, args.valid, args.test, n, r))


if __name__ == "__main__":
    main()
