"""VQ-VAE hyper-parameter presets (values of /root/reference models/utils/config.py:4-80) as plain
namespaces -- no argparse, so importing this never touches sys.argv."""
from types import SimpleNamespace


def _args(n_embed, zquant_dim, in_dim, face_quan_num):
    return SimpleNamespace(vqvae_pretrained_path="", n_embed=n_embed, zquant_dim=zquant_dim, in_dim=in_dim,
                           hidden_size=1024, neg=0.2, quant_factor=0, INaffine=False, num_hidden_layers=6,
                           num_attention_heads=8, intermediate_size=1536, face_quan_num=face_quan_num)


def vq_vae_args():
    return _args(256 * 7, 64, 5023 * 3, 8)


def origin_vq_vae_args():
    return _args(256, 64, 5023 * 3, 8)


def biwi_vq_vae_args():
    return _args(256, 128, 70110, 8)


def vocaset_vq_vae_args():
    return _args(256, 64, 15069, 16)
