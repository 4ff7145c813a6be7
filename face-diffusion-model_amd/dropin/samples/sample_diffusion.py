"""Test-set style sampler -- the loop of /root/reference samples/sample_diffusion_{vocaset,mead,biwi}.py
(sample_step :59-88 / :67-86 / :60-78) over any iterable of (audio, template, one_hots, file_name); without
the datasets (not available here) it runs on seeded synthetic clips.

    python sample_diffusion_vocaset.py --clips 2 --seconds 2 --out result/        (DDIM 100, every training-subject style)
    python sample_diffusion_mead.py / sample_diffusion_biwi.py                     (full DDPM + EVQ / DDIM 50)
    python sample_diffusion.py --dataset vocaset --clips 2 --seconds 2 --ddim_steps 100 --out result/

File names are the reference's: VOCASET writes `<file>_condition_<conditioning subject>.npy` (samples/sample_diffusion_vocaset.py:61-62,
86-88 -- the name computer_metrix.py:71,171-174 and fdm_amd.metrics read back), 3D-MEAD and BIWI write `<file>.npy`
(sample_diffusion_mead.py:86, sample_diffusion_biwi.py:78).
"""
import argparse
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _path  # noqa: F401,E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

from fdm_amd import pipeline, presets  # noqa: E402


# the conditioning subjects a style one-hot stands for, in one-hot order (data of samples/sample_diffusion_vocaset.py:61: the
# VOCASET training subjects); the other presets' samplers do not put the condition in the file name
CONDITION_SUBJECTS = {
    "vocaset": ("FaceTalk_170728_03272_TA FaceTalk_170904_00128_TA FaceTalk_170725_00137_TA FaceTalk_170915_00223_TA "
                "FaceTalk_170811_03274_TA FaceTalk_170913_03279_TA FaceTalk_170904_03276_TA FaceTalk_170912_03278_TA").split(" "),
}
SHIPPED_DDIM_STEPS = {"vocaset": 100, "biwi": 50, "mead": None}      # sample_diffusion_{vocaset:80,biwi:72,mead:81}.py


def save_name(dataset, file_name, style_index):
    """Basename (no extension) of the prediction the reference's sampler of `dataset` writes for `file_name` under the style
    one-hot `style_index`."""
    stem = file_name[:-4]
    subjects = CONDITION_SUBJECTS.get(dataset)
    return f"{stem}_condition_{subjects[style_index]}" if subjects else stem


def synthetic_loader(p, clips, seconds, seed=0, names=None):
    g = torch.Generator().manual_seed(seed)
    for i in range(clips):
        wav = pipeline.processor_normalize((torch.randn(int(seconds * 16000), generator=g) * 0.1).numpy(), pad_seconds=0)
        name = names[i] if names else f"synthetic_{i:03d}.wav"
        yield torch.from_numpy(wav).unsqueeze(0), torch.zeros(1, p.V3), torch.eye(p.n_style).unsqueeze(0), name


@torch.no_grad()
def sample_step(loader, dev, diffusion, autoencoder, save_folder, p, ddim_steps, emotion=4, all_styles=False, batched=True, dataset=None):
    """The reference loops the style one-hots of a clip through one B = 1 sampling call each, with the same audio
    (samples/sample_diffusion_vocaset.py:71-83).  Here all styles of a clip are ONE call (condition-batched step program:
    the audio encoder and audio tables run once per clip); batched=False keeps the sequential loop (same files, bit-identical
    DDIM results)."""
    os.makedirs(save_folder, exist_ok=True)
    dataset = dataset or p.name
    for audio, template, one_hot_all, file_name in loader:
        styles = list(range(one_hot_all.shape[1])) if all_styles else [0]
        emo = torch.eye(p.n_emo)[emotion:emotion + 1] if p.n_emo else None
        steps = None if p.n_emo else ddim_steps
        if batched and len(styles) > 1:
            ids = one_hot_all[0, styles, :]                              # [S, n_style]: one call for the whole style loop
            outs, _ = pipeline.animate(diffusion, autoencoder, audio, template, ids, emo, ddim_steps=steps, device=dev)
            outs = [outs[i:i + 1] for i in range(len(styles))]
        else:
            outs = [pipeline.animate(diffusion, autoencoder, audio, template, one_hot_all[:, it, :], emo, ddim_steps=steps,
                                     device=dev)[0] for it in styles]    # samples/sample_diffusion_vocaset.py:71
        for it, out in zip(styles, outs):
            dst = os.path.join(save_folder, save_name(dataset, file_name, it))
            np.save(dst, out.detach().cpu().numpy())
            print(f"saved {dst}.npy {tuple(out.shape)}")


@torch.no_grad()
def sample_batched(loader, dev, diffusion, autoencoder, save_folder, p, ddim_steps, emotion=4, max_batch=8, dataset=None):
    """The same files as sample_step (style 0 of every clip), but clips of the loader -- whatever their durations -- go through
    ONE sampling call per `max_batch` clips (pipeline.animate_many: exact, the denoiser's attention is causal) instead of the
    reference's one B = 1 call per clip (samples/sample_diffusion_vocaset.py:51: batch size 1)."""
    os.makedirs(save_folder, exist_ok=True)
    items = list(loader)
    audios = [a[0].numpy() for a, _, _, _ in items]
    tmpl = [t for _, t, _, _ in items]
    ids = [oh[:, 0, :] for _, _, oh, _ in items]
    emo = [torch.eye(p.n_emo)[emotion:emotion + 1]] * len(items) if p.n_emo else None
    verts, _ = pipeline.animate_many(diffusion, autoencoder, audios, tmpl, ids, emo, ddim_steps=None if p.n_emo else ddim_steps,
                                     device=dev, max_batch=max_batch)
    for (_, _, _, file_name), out in zip(items, verts):
        dst = os.path.join(save_folder, save_name(dataset or p.name, file_name, 0))
        np.save(dst, out.detach().cpu().numpy())
        print(f"saved {dst}.npy {tuple(out.shape)}")


def main(dataset=None, argv=None):
    """CLI shared by sample_diffusion_{vocaset,mead,biwi}.py (dataset fixed, the reference's shipped schedule and its style loop by
    default) and by this file (--dataset)."""
    ap = argparse.ArgumentParser()
    if dataset is None:
        ap.add_argument("--dataset", default="vocaset", choices=["vocaset", "mead", "biwi"])
    ap.add_argument("--clips", type=int, default=1)
    ap.add_argument("--seconds", type=float, default=2.0)
    ap.add_argument("--ddim_steps", type=int, default=None, help="default: the shipped sampler's (VOCASET DDIM 100, BIWI DDIM 50, MEAD full DDPM)")
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--out", default="result")
    ap.add_argument("--stage1_model_path", default="")
    ap.add_argument("--stage2_model_path", default="")
    ap.add_argument("--all_styles", action="store_true", help="every style one-hot of each clip (the reference's VOCASET loop), as one batched call")
    ap.add_argument("--sequential", action="store_true", help="with --all_styles: one B = 1 call per style, as the reference does")
    ap.add_argument("--batch", type=int, default=1, help="clips per sampling call (clips of different durations batch exactly)")
    a = ap.parse_args(argv)
    named = dataset is not None
    dataset = dataset or a.dataset
    p = presets.get(dataset)
    steps = a.ddim_steps if a.ddim_steps is not None else (SHIPPED_DDIM_STEPS[dataset] or 100)
    all_styles = a.all_styles or (named and dataset == "vocaset")      # samples/sample_diffusion_vocaset.py:71 loops every one-hot
    # one clip, one condition per sampling call (the reference's bs = 1 loop): the step program's single-clip setting
    single = a.batch <= 1 and (not all_styles or a.sequential)
    diffusion, ae = pipeline.build_models(dataset, None, a.device, a.stage1_model_path, a.stage2_model_path, single_clip=single)
    if a.batch > 1 and not all_styles:
        sample_batched(synthetic_loader(p, a.clips, a.seconds), a.device, diffusion, ae, a.out, p, steps, max_batch=a.batch, dataset=dataset)
    else:
        sample_step(synthetic_loader(p, a.clips, a.seconds), a.device, diffusion, ae, a.out, p, steps,
                    all_styles=all_styles, batched=not a.sequential, dataset=dataset)


if __name__ == "__main__":
    main()
