"""A small synthetic preprocessed-feature tree in the reference's on-disk format (dataset.py:41-70,95-109): `<txt>` lines
`basename|speaker|{ARPAbet phones}|raw text`, `speakers.json`, `<kind>/<speaker>-<kind>-<basename>.npy`.  Shared by
tests/test_data.py and tests/golden/make_collate_golden.py so both sides read byte-identical files."""
import json
import os

import numpy as np

SPEAKERS = {"spkA": 12, "spkB": 9, "spkC": 3}
PHONES = ["AH0", "B", "K", "S", "T", "IY1", "N", "D", "M", "EH1", "sp", "L", "R", "AE1", "Z", "OW1"]


def write_tree(root, n_mel=32, seed=0):
    g = np.random.RandomState(seed)
    for kind in ("mel", "pitch", "energy", "duration"):
        os.makedirs(os.path.join(root, kind), exist_ok=True)
    lines = []
    for spk, n in SPEAKERS.items():
        for u in range(n):
            base = f"{spk}_utt{u:02d}"
            S = int(g.randint(5, 13))
            dur = g.randint(1, 6, size=S)
            T = int(dur.sum())
            np.save(os.path.join(root, "mel", f"{spk}-mel-{base}.npy"), g.standard_normal((T, n_mel)).astype(np.float32))
            np.save(os.path.join(root, "pitch", f"{spk}-pitch-{base}.npy"), g.standard_normal(S))          # float64 on disk, as pyworld writes
            np.save(os.path.join(root, "energy", f"{spk}-energy-{base}.npy"), g.standard_normal(S).astype(np.float32))
            np.save(os.path.join(root, "duration", f"{spk}-duration-{base}.npy"), dur)
            phones = " ".join(PHONES[int(x)] for x in g.randint(0, len(PHONES), size=S))
            lines.append(f"{base}|{spk}|{{{phones}}}|raw text of {base}")
    # partial-utterance reference mels of the dvec / encoder speaker modes (dataset.py:83-91): (n_partials, frames, 40); written
    # from their own generator so that the files above stay byte-identical to the ones the older fixtures were made from
    os.makedirs(os.path.join(root, "spk_ref_mel_slices"), exist_ok=True)
    g2 = np.random.RandomState(seed + 1)
    for ln in lines:
        base, spk = ln.split("|")[:2]
        np.save(os.path.join(root, "spk_ref_mel_slices", f"{spk}-mel-{base}.npy"), g2.standard_normal((int(g2.randint(1, 4)), 6, 40)).astype(np.float32))
    with open(os.path.join(root, "train.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    with open(os.path.join(root, "speakers.json"), "w") as f:
        json.dump({s: i + 3 for i, s in enumerate(SPEAKERS)}, f)
    return lines
