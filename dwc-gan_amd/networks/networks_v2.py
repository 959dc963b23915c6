"""MI355X-native generator behind the reference's ``networks.networks_v2`` API.

``AdaINGen_v2`` keeps the constructor, ``encode`` / ``encode_txt`` / ``decode`` methods, attribute
names and ``state_dict`` keys of reference networks/networks_v2.py:9-95; the conv / norm /
upsample stacks run on libdwcgan_hip.so through ``hipdwc.ops``.  The text encoder's packed
bi-LSTM runs its sequential part on the HIP recurrent kernels (``ops.lstm_bidir``; embedding lookup
and the input / weight-gradient GEMMs are library calls), with the reference's batch-mixing
``view`` reproduced on purpose.
"""
import numpy as np
import torch
from torch import nn

from hipdwc import host, ops
from .networks import ContentEncoder, MLP, Conv2dBlock, ResBlocks, AdaptiveInstanceNorm2d, Decoder  # noqa: F401


class HeadList(list):
    """The reference's per-attribute head list ([mu_0 .. mu_{K-1}], each [B, c_dim]) that also remembers the ONE [B, K*c_dim]
    tensor its entries are column slices of (``flat``).  Callers inside this package read ``flat`` (``flat_heads``) instead
    of concatenating / stacking the K slices again: one autograd edge instead of K slice / cat / add nodes per use (the K-way
    lists cost ~250 of the ~770 stock-torch launches of a training step, r03 kernel trace).  To the reference's callers it is
    an ordinary list."""
    flat = None

    @classmethod
    def of(cls, flat, k):
        out = cls(flat.reshape(flat.shape[0], k, -1).unbind(1))
        out.flat = flat
        return out

    def rows(self, a, b):
        """The same heads for samples a..b of the batch."""
        return HeadList.of(self.flat[a:b], len(self))


def flat_heads(heads):
    """[B, K*c_dim] of a head list (its ``flat`` tensor when it has one, else the concatenation the reference writes)."""
    if torch.is_tensor(heads):
        return heads
    flat = getattr(heads, "flat", None)
    return flat if flat is not None else torch.cat(list(heads), dim=1)


class StyleEncoder(nn.Module):
    """Image -> per-attribute (mu, logvar) heads (reference networks_v2.py:98-141): 7x7 stem,
    n_downsample stride-2 convs without norm, global average pool, optional 2-layer mapping
    with dropout 0.1, then num_class pairs of Linear(dim -> c_dim)."""

    def __init__(self, n_downsample, input_dim, dim, norm, activ, pad_type, c_dim, num_class, use_map=False):
        super().__init__()
        self.num_class, self.use_map, self.c_dim = num_class, use_map, c_dim
        convs = [Conv2dBlock(input_dim, dim, 7, 1, 3, norm=norm, activation=activ, pad_type=pad_type)]
        for i in range(n_downsample):
            nxt = dim * 2 if i < 2 else dim
            convs.append(Conv2dBlock(dim, nxt, 4, 2, 1, norm=norm, activation=activ, pad_type=pad_type))
            dim = nxt
        convs.append(nn.AdaptiveAvgPool2d(1))
        self.model = nn.Sequential(*convs)
        if use_map:
            self.mapping = nn.Sequential(nn.Linear(dim, dim), nn.ReLU(inplace=True), nn.Dropout(p=0.1),
                                         nn.Linear(dim, dim), nn.ReLU(inplace=True))
        self.fcs, self.fcvars = nn.ModuleList(), nn.ModuleList()
        for _ in range(num_class):           # interleaved creation order matters for seeded init
            self.fcs.append(nn.Linear(dim, c_dim))
            self.fcvars.append(nn.Linear(dim, c_dim))
        self.output_dim = dim

    def forward(self, x, drop_mask=None):
        """``drop_mask``: optional pre-drawn scaled keep-mask for the mapping dropout (lets a caller
        that batches several encodes draw the masks in the reference's order)."""
        h = ops.pack_image(x) if x.shape[1] < 4 else x
        for blk in list(self.model)[:-1]:
            h = blk(h)
        f = h.float().mean(dim=(2, 3))                           # global average pool over <=4x4 pixels (fp32 from here on)
        if self.use_map:
            f = ops.linear(f, self.mapping[0].weight, self.mapping[0].bias, "relu")
            if self.training and self.mapping[2].p > 0:
                if drop_mask is None:
                    drop_mask = host.noise().dropout_mask(f.shape, self.mapping[2].p, f.device)
                f = f * drop_mask
            f = ops.linear(f, self.mapping[3].weight, self.mapping[3].bias, "relu")
        # the 2*num_class heads share their input: one [2*num_class*c_dim, dim] product
        heads_w = [m.weight for m in self.fcs] + [m.weight for m in self.fcvars]
        w = ops.cat_params(heads_w)
        b = ops.cat_params([m.bias for m in self.fcs] + [m.bias for m in self.fcvars])
        out = ops.linear(f, w, b, owner=tuple(heads_w))     # (prepared-weight cache keyed on the 16 parameters, not on the fresh view)
        k, c = self.num_class, self.c_dim
        return HeadList.of(out[:, :k * c], k), HeadList.of(out[:, k * c:2 * k * c], k)


def _packed_index(lens_sorted, t_max, bsz, device):
    """Row of the PackedSequence data tensor that holds (t, b), for lengths sorted descending; slots past a sample's
    length point at row 0 (their values are zero anyway).  [t_max*bsz] int64."""
    idx = np.zeros((t_max, bsz), dtype=np.int64)
    off = 0
    for t in range(t_max):
        n = sum(1 for v in lens_sorted if v > t)
        idx[t, :n] = off + np.arange(n)
        off += n
    return torch.from_numpy(idx.reshape(-1)).to(device, non_blocking=True)


class _Permute(torch.autograd.Function):
    """``x.index_select(dim, perm)`` for a PERMUTATION ``perm`` with inverse ``inv``: the backward is the gather by ``inv`` instead of
    autograd's ``index_add_`` (a 64-thread kernel that took 109 us per step on the [B, 64] style code)."""

    @staticmethod
    def forward(ctx, x, dim, perm, inv):
        ctx.dim, ctx.inv = dim, inv
        return x.index_select(dim, perm)

    @staticmethod
    def backward(ctx, g):
        return g.index_select(ctx.dim, ctx.inv), None, None, None


class TxtEncoder(nn.Module):
    """Command text + current style -> per-attribute (mu, logvar) (reference networks_v2.py:171-254)."""

    def __init__(self, vocab, embed_dim=512, hidden_size=512, c_dim=8, num_class=8, num_layers=1, dropout_in=0.1,
                 dropout_out=0.1, bidirectional=True, pretrained_embed=None):
        super().__init__()
        self.vocab, self.embed_dim, self.hidden_size = vocab, embed_dim, hidden_size
        self.num_layers, self.dropout_in, self.dropout_out = num_layers, dropout_in, dropout_out
        self.bidirectional, self.num_class, self.style_dim = bidirectional, num_class, c_dim * num_class
        self.embed_tokens = nn.Embedding(vocab.size, embed_dim, vocab.padding_idx)
        if pretrained_embed is not None:
            table = np.zeros((vocab.size, embed_dim))
            for i, word in enumerate(vocab.itos):
                table[i] = pretrained_embed[word] if word in pretrained_embed else \
                    np.random.normal(scale=0.6, size=(embed_dim,))
            self.embed_tokens.load_state_dict({"weight": torch.from_numpy(table)})
            self.embed_tokens.weight.requires_grad = False
        self.lstm = nn.LSTM(input_size=embed_dim + self.style_dim, hidden_size=hidden_size, num_layers=num_layers,
                            dropout=dropout_out if num_layers > 1 else 0.0, bidirectional=bidirectional)
        feat = hidden_size * num_layers * (4 if bidirectional else 2)
        self.fcs, self.fcvars = nn.ModuleList(), nn.ModuleList()
        for _ in range(num_class):
            self.fcs.append(nn.Linear(feat, c_dim))
            self.fcvars.append(nn.Linear(feat, c_dim))

    def forward(self, style_ord, src_tokens, src_lengths):
        noise = host.noise()
        tokens = src_tokens.transpose(1, 0)
        seq_len, bsz = tokens.shape
        lens_host = src_lengths.detach().to("cpu")               # kept on the host: no device sync per call
        lens_sorted, order_host = torch.sort(lens_host, descending=True)
        # order, its inverse, the sorted lengths and the last-token index travel in ONE pinned staging block, asynchronously (r06: four
        # separate copies from pageable memory made the host wait for the stream four times per call -- with torch.normal's check that was
        # why the host could never run ahead of the GPU, benchmarks/host_vs_gpu.py)
        if tokens.is_cuda and ops.PINNED_STAGE:
            stage = torch.empty(4, bsz, dtype=torch.int64).pin_memory()
            stage[0], stage[1], stage[2], stage[3] = order_host, torch.sort(order_host)[1], lens_sorted, lens_sorted - 1
            on_dev = stage.to(tokens.device, non_blocking=True)
            order, unsort, lens_dev, last = on_dev[0], on_dev[1], on_dev[2].to(torch.int32), on_dev[3]
        else:
            order, unsort = order_host.to(tokens.device), torch.sort(order_host)[1].to(tokens.device)
            lens_dev, last = lens_sorted.to(torch.int32).to(tokens.device), (lens_sorted - 1).to(tokens.device)
        emb = self.embed_tokens(tokens.index_select(1, order))
        emb = noise.dropout(emb, self.dropout_in, self.training)
        sty = _Permute.apply(style_ord, 0, order, unsort)
        # The packed bi-LSTM on the HIP recurrent kernels (hipdwc.ops.lstm_bidir): padded [T,B,*] tensors with per-sample
        # lengths instead of a PackedSequence; T = longest sequence of the batch, as pack_padded_sequence would cut it.
        lens_list = lens_sorted.tolist()
        t_max = int(lens_list[0])
        data = torch.cat([emb, sty.expand(seq_len, -1, -1)], -1)[:t_max]
        cols = torch.arange(bsz, device=tokens.device)
        suffixes = ("", "_reverse") if self.bidirectional else ("",)
        if not self.bidirectional:
            raise NotImplementedError("the HIP text encoder is bidirectional (the reference's only configuration)")
        hs, cs = [], []
        for l in range(self.num_layers):
            par = {n: ops.cat_params([getattr(self.lstm, "%s_l%d%s" % (n, l, suf)) for suf in suffixes], stack=True)
                   for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")}
            out, cell = ops.lstm_bidir(data, lens_dev, par["weight_ih"], par["weight_hh"], par["bias_ih"], par["bias_hh"],
                                       owners=tuple(getattr(self.lstm, "weight_ih_l%d%s" % (l, suf)) for suf in suffixes))
            # final states: forward direction at each sample's last token, reverse direction at t = 0
            hs += [out[0][last, cols], out[1][0]]
            cs += [cell[0][last, cols], cell[1][0]]
            data = torch.cat([out[0], out[1]], -1)
            if l + 1 < self.num_layers and self.training and self.dropout_out > 0:
                # nn.LSTM(dropout=p) between the layers.  In parity mode the mask is drawn in PACKED order (the shape
                # and stream consumption of a stock nn.LSTM on CPU) and scattered onto the padded layout.
                if getattr(noise, "align_stream", False):
                    mask = noise.dropout_mask((int(sum(lens_list)), data.shape[2]), self.dropout_out, data.device)
                    data = data * mask.index_select(0, _packed_index(lens_list, t_max, bsz, data.device)).view(t_max, bsz, -1)
                else:
                    data = noise.dropout(data, self.dropout_out, True)
        h_n, c_n = torch.stack(hs), torch.stack(cs)
        if self.training and self.dropout_out > 0 and noise.align_stream:
            # the reference drops out the (unused) padded memory here and thereby advances the
            # random stream (reference networks_v2.py:235-236); parity mode keeps the stream aligned
            noise.dropout(data, self.dropout_out, True)
        h_n = h_n.view(self.num_layers, 2, bsz, -1).transpose(1, 2).reshape(self.num_layers, bsz, -1)
        c_n = c_n.view(self.num_layers, 2, bsz, -1).transpose(1, 2).reshape(self.num_layers, bsz, -1)
        h_n, c_n = _Permute.apply(h_n, 1, unsort, order), _Permute.apply(c_n, 1, unsort, order)
        # reference networks_v2.py:249: concatenating along the BATCH axis and then viewing as
        # (batch, -1) interleaves samples of the local batch; reproduced, not fixed
        feat = torch.cat([h_n, c_n], dim=1).view(bsz, -1)
        # the 2*num_class heads read the same feature row: one [2*num_class*c_dim, feat] product
        w = ops.cat_params([m.weight for m in self.fcs] + [m.weight for m in self.fcvars])
        b = ops.cat_params([m.bias for m in self.fcs] + [m.bias for m in self.fcvars])
        if feat.is_cuda and ops.LSTM_GEMM and ops.gemm_ok(feat.shape[1], w.shape[0]):
            out = ops.linear_any(feat, w, b, owner=tuple(m.weight for m in self.fcs) + tuple(m.weight for m in self.fcvars))
        else:
            out = torch.nn.functional.linear(feat, w, b)
        k, c = self.num_class, self.style_dim // self.num_class
        return HeadList.of(out[:, :k * c], k), HeadList.of(out[:, k * c:2 * k * c], k)


class AdaINGen_v2(nn.Module):
    """Style/content auto-encoder with AdaIN injection (reference networks_v2.py:9-95)."""

    def __init__(self, input_dim, vocab, params, pretrained_embed=None):
        super().__init__()
        p = params
        style_dim = p["c_dim"] * p["num_cls"]
        self.enc_style = StyleEncoder(p["style_downsample"], input_dim, p["dim"], norm="none", activ=p["activ"],
                                      pad_type=p["pad_type"], c_dim=p["c_dim"], num_class=p["num_cls"],
                                      use_map=p["use_map"])
        self.enc_content = ContentEncoder(p["content_downsample"], p["n_res"], input_dim, p["dim"], "in", p["activ"],
                                          pad_type=p["pad_type"])
        self.dec = Decoder(p["content_downsample"], p["n_res"], self.enc_content.output_dim, input_dim,
                           res_norm="adain", activ=p["activ"], pad_type=p["pad_type"],
                           use_attention=p["use_attention"])
        self.enc_txt = TxtEncoder(vocab, p["embed_dim"], p["hidden_size"], p["c_dim"], p["num_cls"], p["num_layers"],
                                  p["dropout_in"], p["dropout_out"], pretrained_embed=pretrained_embed)
        self.mlp = MLP(style_dim, self.get_num_adain_params(self.dec), p["mlp_dim"], 3, norm="none", activ=p["activ"])

    # -- reference API -------------------------------------------------------------------
    def forward(self, images):
        content, mus, _ = self.encode(images)
        return self.decode(content, mus)

    def encode(self, images, drop_mask=None):
        x = ops.pack_image(images)
        mus, logvar = self.enc_style(x, drop_mask=drop_mask)
        return self.enc_content(x), mus, logvar

    def draw_encode_mask(self, batch, device):
        """The dropout mask one ``encode`` of ``batch`` images would draw (None when nothing is drawn)."""
        es = self.enc_style
        if not (es.use_map and es.training and es.mapping[2].p > 0):
            return None
        return host.noise().dropout_mask((batch, es.mapping[0].out_features), es.mapping[2].p, device)

    def encode_txt(self, style_ord, txt_org2trg, txt_lens):
        return self.enc_txt(style_ord, txt_org2trg, txt_lens)

    def decode(self, content, style):
        self.assign_adain_params(self.mlp(style), self.dec)
        return self.dec(content)

    # -- internal 4-plane image API used by the solver ------------------------------------
    def decode_nhwc4(self, content, style, attention_used=True, groups=1):
        """As decode(), returning the fused NHWC4 head buffer (planes 0-2 image, plane 3 attention).  ``groups`` > 1: ``content`` is ONE
        copy of a batch that is decoded with ``groups`` styles per sample, ``style`` has groups * B rows (group-major)."""
        self.assign_adain_params(self.mlp(style), self.dec)
        return self.dec.forward_nhwc4(content, attention_used=attention_used, groups=groups)

    def assign_adain_params(self, adain_params, model):
        """Hand each AdaIN layer, in module order, its slice of the MLP output: first C columns ->
        bias (shift), next C -> weight (scale), both flattened to B*C (reference networks_v2.py:78-87)."""
        layers = [m for m in model.modules() if m.__class__.__name__ == "AdaptiveInstanceNorm2d"]
        widths = {m.num_features for m in layers}
        if len(widths) == 1 and adain_params.shape[1] >= 2 * len(layers) * layers[0].num_features:
            # equal widths (the shipped decoder): ONE transposing copy [B, L*2, C] -> [L*2, B*C] whose rows are the
            # flattened per-layer vectors, handed out by unbind (one autograd node instead of 2L slices + 2L clones)
            c, n = layers[0].num_features, 2 * len(layers)
            rows = adain_params[:, :n * c].reshape(adain_params.shape[0], n, c).transpose(0, 1).reshape(n, -1).unbind(0)
            for i, m in enumerate(layers):
                m.bias, m.weight = rows[2 * i], rows[2 * i + 1]
            return
        col = 0
        for m in layers:
            c = m.num_features
            m.bias = adain_params[:, col:col + c].contiguous().view(-1)
            m.weight = adain_params[:, col + c:col + 2 * c].contiguous().view(-1)
            col += 2 * c

    def get_num_adain_params(self, model):
        return sum(2 * m.num_features for m in model.modules() if m.__class__.__name__ == "AdaptiveInstanceNorm2d")
