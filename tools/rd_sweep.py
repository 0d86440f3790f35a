#!/usr/bin/env python3
"""RD sweep over a directory of reference-format checkpoints (the job of /root/reference/eval_CLC.py:main) on the MI355X engine.

  python tools/rd_sweep.py --models_dir DIR --data IMAGES_DIR [--ref_dir REFS_DIR] [--n_refs 3] [--model clc|tcm] [--out DIR]

DIR holds <tag>_<lambda>/<lambda>checkpoint_best.pth.tar (eval_CLC.py:183-204).  IMAGES_DIR: image files; references for image
`name.png` are REFS_DIR/name/*.png (first n_refs, as KodakDataset does, eval_CLC.py:27-131), resized to the image.  With
--synthetic N the sweep runs on N seeded synthetic images (no dataset is reachable in the build environment)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--models_dir", required=True)
    ap.add_argument("--data")
    ap.add_argument("--ref_dir")
    ap.add_argument("--n_refs", type=int, default=3)
    ap.add_argument("--model", default="clc", choices=["clc", "tcm"])
    ap.add_argument("--N", type=int, default=64)
    ap.add_argument("--synthetic", type=int, default=0)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import numpy as np
    import torch

    from clc_amd import eval as ev
    from clc_amd import models
    from clc_amd.recipe import synthetic_image

    def load(path):
        from PIL import Image

        return torch.from_numpy(np.asarray(Image.open(path).convert("RGB"), dtype=np.float32).transpose(2, 0, 1) / 255.0)

    if args.synthetic:
        samples = [(synthetic_image(1, 512, 768, 100 + i, smooth=True)[0], [synthetic_image(1, 512, 768, 500 + 10 * i + j, smooth=True)[0] for j in range(args.n_refs)])
                   for i in range(args.synthetic)]
    else:
        samples = []
        for f in sorted(os.listdir(args.data)):
            if not f.lower().endswith((".png", ".jpg", ".jpeg", ".bmp")):
                continue
            refs = []
            if args.ref_dir:
                rd = os.path.join(args.ref_dir, os.path.splitext(f)[0])
                refs = [load(os.path.join(rd, r)) for r in sorted(os.listdir(rd))[: args.n_refs]] if os.path.isdir(rd) else []
            samples.append((load(os.path.join(args.data, f)), refs))
    make = (lambda: models.CLC(N=args.N, num_ref_frames=args.n_refs)) if args.model == "clc" else (lambda: models.TCM(N=args.N))
    cps = ev.find_checkpoints(args.models_dir)
    print(f"{len(cps)} checkpoints, {len(samples)} images")
    results, csv_path = ev.rd_sweep(make, cps, samples, args.out or os.path.join(args.models_dir, "rd_curve_results"))
    for r in results:
        print(f"{r['checkpoint']}: {r['bitrate']:.4f} bpp, {r['psnr']:.2f} dB, {r['time']:.4f} s/image")
    print("wrote", csv_path)


if __name__ == "__main__":
    main()
