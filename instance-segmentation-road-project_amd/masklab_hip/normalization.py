"""GroupNormalization -- drop-in for reference engine/normalization.py (the Keras layer),
running the chunk-norm HIP kernel (csrc/groupnorm.hip).

The reference layer with axis=-1 on NHWC does NOT group channels: it reshapes [N,H,W,C]
row-major to [N,G,H,W,C/G] and normalises over axes (2,3,4) (normalization.py:123-143), i.e.
each sample's flat H*W*C vector is cut into G contiguous chunks; gamma/beta are indexed
g*(C/G) + (c mod C/G) (:151-156).  That exact behaviour is reproduced.
"""
import numpy as np

from . import ops
from .keras_like import Layer


class GroupNormalization(Layer):
    def __init__(self, groups=32, axis=-1, epsilon=1e-5, center=True, scale=True,
                 beta_initializer="zeros", gamma_initializer="ones", beta_regularizer=None,
                 gamma_regularizer=None, beta_constraint=None, gamma_constraint=None, **kwargs):
        super().__init__(**kwargs)
        if axis not in (-1, 3):
            raise NotImplementedError("GroupNormalization: only axis=-1 (NHWC) is on the hot path")
        self.supports_masking = True
        self.groups = groups
        self.axis = axis
        self.epsilon = epsilon
        self.center = center
        self.scale = scale
        self.beta_initializer = beta_initializer
        self.gamma_initializer = gamma_initializer
        self.beta_regularizer = beta_regularizer
        self.gamma_regularizer = gamma_regularizer
        self.beta_constraint = beta_constraint
        self.gamma_constraint = gamma_constraint
        self.gamma = self.beta = None

    def build(self, input_shape):
        dim = input_shape[self.axis]
        # same checks / messages as reference normalization.py:78-92
        if dim is None:
            raise ValueError('Axis ' + str(self.axis) + ' of input tensor should have a defined dimension '
                             'but the layer received an input with shape ' + str(input_shape) + '.')
        if dim < self.groups:
            raise ValueError('Number of groups (' + str(self.groups) + ') cannot be '
                             'more than the number of channels (' + str(dim) + ').')
        if dim % self.groups != 0:
            raise ValueError('Number of groups (' + str(self.groups) + ') must be a '
                             'multiple of the number of channels (' + str(dim) + ').')
        self.dim = int(dim)
        # synthetic-weight init draws non-trivial gamma/beta so parity tests exercise the indexing
        if self.scale:
            self.add_weight("gamma", (dim,), "uniform", low=0.5, high=1.5)
        if self.center:
            self.add_weight("beta", (dim,), "normal", stddev=0.1)
        self.built = True
        return input_shape

    def _load_own(self, weights, device):
        import torch
        self.gamma = torch.from_numpy(self._get(weights, "gamma")).to(device) if self.scale else None
        self.beta = torch.from_numpy(self._get(weights, "beta")).to(device) if self.center else None

    def call(self, inputs, fuse_relu=False, inplace=False, **kwargs):
        if not self.built:
            self.build(tuple(inputs.shape))
        if (self.scale and self.gamma is None) or (self.center and self.beta is None):
            raise RuntimeError(f"layer '{self.name}' has no weights loaded")
        return ops.groupnorm_chunk(inputs, self.gamma, self.beta, self.groups, self.epsilon, relu=fuse_relu,
                                   out=inputs if inplace else None)

    @staticmethod
    def call_multi(layers, inputs, inplace=False, lives=None, partials=None):
        """The same layers applied to their own inputs in ONE launch pair (the un-shared towers normalise five pyramid
        levels at every depth, reference engine/layers/detection.py:124,194): results identical to calling each.
        lives: per input None or (device int32 [1], slots per image) -- a fixed-capacity RoI batch whose samples past
        max(1, live) per image do not exist and are skipped."""
        probs = []
        lives = lives if lives is not None else [None] * len(inputs)
        partials = partials if partials is not None else [None] * len(inputs)   # per input None or (float64 pairs, per chunk)
        for layer, x, live, part in zip(layers, inputs, lives, partials):
            if not layer.built:
                layer.build(tuple(x.shape))
            if (layer.scale and layer.gamma is None) or (layer.center and layer.beta is None):
                raise RuntimeError(f"layer '{layer.name}' has no weights loaded")
            hwc = x.numel() // x.shape[0]
            vw = 16 // x.element_size()                             # elements per 16-byte access: 4 floats / 8 halves
            if (hwc // layer.groups) % vw or x.shape[-1] % vw:      # the multi launch takes vectorisable problems only
                if any(lv is not None for lv in lives):
                    raise NotImplementedError("GroupNormalization: fixed-capacity batches need vectorisable chunks")
                return [l(x_, inplace=inplace) for l, x_ in zip(layers, inputs)]
            probs.append(dict(x=x, gamma=layer.gamma, beta=layer.beta, groups=layer.groups, eps=layer.epsilon,
                              out=x if inplace else None, live=live, partials=part))
        return ops.groupnorm_chunk_multi(probs)

    def get_config(self):
        config = {
            'groups': self.groups, 'axis': self.axis, 'epsilon': self.epsilon, 'center': self.center,
            'scale': self.scale, 'beta_initializer': self.beta_initializer,
            'gamma_initializer': self.gamma_initializer, 'beta_regularizer': self.beta_regularizer,
            'gamma_regularizer': self.gamma_regularizer, 'beta_constraint': self.beta_constraint,
            'gamma_constraint': self.gamma_constraint,
        }
        base = super().get_config()
        return dict(list(base.items()) + list(config.items()))

    def compute_output_shape(self, input_shape):
        return input_shape


__all__ = ["GroupNormalization"]
