"""Fixture generator (run in the build container, where /root/reference exists): the parameter / buffer names of the DETR-101 model
the reference loads its checkpoints into - its own data file ``datasets/vg_scene_graph_annot/detr101_key_after.txt`` - copied as
``tests/golden/detr101_keys.txt``.  Names only: this pins the state-dict contract of ``scene_graph_commonsense_amd/detr.py``."""
import os

SRC = "/root/reference/datasets/vg_scene_graph_annot/detr101_key_after.txt"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "detr101_keys.txt")

if __name__ == "__main__":
    names = [l.strip() for l in open(SRC) if l.strip()]
    with open(DST, "w") as f:
        f.write("\n".join(names) + "\n")
    print(len(names), "names ->", DST)
