"""Post-process of the decoded image: masked per-channel histogram matching.

Mirror of GeoDiffuser/utils/image_processing.py:24-77 (``masked_histogram_matching``; same name / arguments): for each
colour channel the 256-bin histogram of the source inside ``mask_source`` is mapped onto the histogram of the template
inside ``mask`` through their cumulative distributions.  Runs once per edit on the host on a 512x512x3 uint8 image
(a "next" row of the scope table — the device version is a 256-bin integer kernel); float64 result like the reference.
"""
from __future__ import annotations

import numpy as np


def _match_cumulative_cdf(source, template, mask=None, mask_source=None):
    if mask is None:
        mask = np.ones_like(source)
    if mask_source is None:
        mask_source = mask
    src_sel = source[mask_source > 0.5].reshape(-1)
    tmpl_sel = template[mask > 0.5].reshape(-1)
    src_counts = np.bincount(src_sel, minlength=256)
    tmpl_counts = np.bincount(tmpl_sel, minlength=256)
    tmpl_values = np.linspace(0, 255, 256).astype("uint8")
    src_quantiles = np.cumsum(src_counts) / src_sel.size
    tmpl_quantiles = np.cumsum(tmpl_counts) / tmpl_sel.size
    lut = np.interp(src_quantiles, tmpl_quantiles, tmpl_values)
    return lut[source.reshape(-1)].reshape(source.shape)


def masked_histogram_matching(source, template, mask=None, mask_source=None):
    return np.stack([_match_cumulative_cdf(source[..., i], template[..., i], mask, mask_source)
                     for i in range(source.shape[-1])], -1)
