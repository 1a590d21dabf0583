"""PriorBoxes -- anchor table, same constructor / `len()` / `.boxes` / `.config` surface as the
reference's engine/prior.py:9-71.  The table is pinned against the reference's own output
(tests/golden/prior_tables.npz)."""
import numpy as np


class PriorBoxes:
    """Default Box Configuration Class"""

    def __init__(self, strides, sizes, pr_scales, pr_ratios):
        as_list = lambda v: v.tolist() if isinstance(v, np.ndarray) else list(v)
        self.strides = as_list(strides)
        self.sizes = as_list(sizes)
        self.pr_scales = as_list(pr_scales)
        self.pr_ratios = as_list(pr_ratios)
        assert len(self.strides) == len(self.sizes), "the number of strides and sizes must match"
        self.setup()
        self.config = {"strides": self.strides, "sizes": self.sizes,
                       "pr_scales": self.pr_scales, "pr_ratios": self.pr_ratios}

    def __len__(self):
        """number of anchors per grid point"""
        return len(self.pr_scales) * len(self.pr_ratios)

    def setup(self):
        # rows (stride, w, h), loops size/stride -> scale -> ratio, np.round (half to even)
        rows = []
        for size, stride in zip(self.sizes, self.strides):
            for wh_size in self.pr_scales:
                for wh_ratio in self.pr_ratios:
                    w = int(np.round(size * wh_size * np.sqrt(wh_ratio)))
                    h = int(np.round(size * wh_size / np.sqrt(wh_ratio)))
                    rows.append((int(stride), w, h))
        self.table = np.asarray(rows, np.int64).reshape(-1, 3)

    @property
    def boxes(self):
        """pandas DataFrame view (columns stride,w,h; index from 1) like the reference's attribute."""
        import pandas as pd
        return pd.DataFrame(self.table, columns=['stride', 'w', 'h'], index=np.arange(1, len(self.table) + 1))

    def get_config(self):
        return self.config

    def anchors(self, height, width, padding='same'):
        """[A,4] int32 (cx,cy,w,h) for an image size -- what PriorLayer.call builds
        (reference engine/layers/detection.py:269-295, without the batch tile)."""
        out = []
        for stride in sorted(set(self.table[:, 0].tolist())):
            rows = self.table[self.table[:, 0] == stride]
            rnd = np.ceil if padding == 'same' else np.floor
            th = int(rnd(height / stride) * stride)
            tw = int(rnd(width / stride) * stride)
            ys = np.arange(stride // 2, th, stride)
            xs = np.arange(stride // 2, tw, stride)
            gx, gy = np.meshgrid(xs, ys)
            lvl = np.empty(gx.shape + (len(rows), 4), np.int32)
            lvl[..., 0] = gx[..., None]
            lvl[..., 1] = gy[..., None]
            lvl[..., 2] = rows[:, 1]
            lvl[..., 3] = rows[:, 2]
            out.append(lvl.reshape(-1, 4))
        return np.concatenate(out, axis=0)
