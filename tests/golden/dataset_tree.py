"""Directory layouts of the reference's dataset readers on five small images: shared by make_golden.py (which runs
the reference readers on it) and tests/test_dataloader.py (which runs the mirror on the same tree)."""
import os


def build_dataset_tree(root, imgs):
    """The directory layouts of the reference's readers, filled with the fixture's five small images (shared with
    tests/test_dataloader.py through import)."""
    import json
    import PIL.Image

    root = str(root)
    save = lambda rel, k: (os.makedirs(os.path.dirname(os.path.join(root, rel)), exist_ok=True),
                           PIL.Image.fromarray(imgs[k]).save(os.path.join(root, rel)))
    tum = "tum/rgbd_dataset_freiburg1_tiny"
    lines = ["# color images", "# file: 'tiny.bag'", "# timestamp filename"]
    for k in range(5):
        save(f"{tum}/rgb/1305031102.{175304 + 33333 * k}.png", k)
        lines.append(f"1305031102.{175304 + 33333 * k} rgb/1305031102.{175304 + 33333 * k}.png")
    open(os.path.join(root, tum, "rgb.txt"), "w").write("\n".join(lines) + "\n")
    for k, n in enumerate((0, 1, 2, 10, 11)):
        save(f"7-scenes/chess/seq-01/frame-{n:06d}.color.png", k)
        save(f"7-scenes/chess/seq-01/frame-{n:06d}.depth.png", k)
    for k, n in enumerate((2, 10, 1)):
        save(f"plain/{n}.png", k)
    eth = "eth3d/tiny"
    lines = []
    for k in range(4):
        save(f"{eth}/rgb/{k}.png", k)
        lines.append(f"{0.5 * k:.6f} rgb/{k}.png")
    open(os.path.join(root, eth, "rgb.txt"), "w").write("\n".join(lines) + "\n")
    open(os.path.join(root, eth, "calibration.txt"), "w").write("60.0 61.0 32.0 24.0\n")
    rep = "Replica/office0"
    for k in range(4):
        save(f"{rep}/results/frame{k:06d}.png", k)
        save(f"{rep}/results/depth{k:06d}.png", k)
    open(os.path.join(root, rep, "traj.txt"), "w").write("".join(f"{0.1 * k} 0 0 0 0 0 0 1\n" for k in range(5)))
    json.dump({"camera": {"fx": 60.0, "fy": 60.0, "cx": 32.0, "cy": 24.0}}, open(os.path.join(root, "Replica/cam_params.json"), "w"))
    return {"TUMDataset": tum, "SevenScenesDataset": "7-scenes/chess", "RGBFiles": "plain", "ETH3DDataset": eth, "ReplicaDataset": rep}
