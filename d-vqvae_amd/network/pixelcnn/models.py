"""Class-conditional GatedPixelCNN prior over the 3x3 grid of code indices.

Mirrors DVQ-VAE/network/pixelcnn/models.py:21-198 (same module tree, 140 state_dict keys
for GatedPixelCNN(512, 512, 15)); forward(x, label) returns the same logits.

generate() draws the grid cell by cell like the reference (models.py:176-198) but does
not re-run the whole network nine times.  The masked convolutions are causal -- the
logits of cell (i,j) depend only on rows < i and on cells (i, <j) -- so every
activation the reference recomputes for a finished cell is identical in all later
passes.  Here each activation is computed once:
  * when a grid row starts, the vertical stack of that whole row (all 15 layers), which
    only sees rows above it, and its 1x1 projection into the horizontal stack;
  * per cell, the horizontal stack at that single position, from the cached
    activations of its left neighbour, then the output head and one sample.
That is one network evaluation per grid instead of nine, with every product a plain
[batch, K] x [K, N] matrix product over the whole batch.  Activations are kept
position-major ([cell][batch][channel]) so each product reads contiguous rows.

Differences from the reference, none visible at its only supported batch size of 1:
the mask-'A' zeroing is applied to a masked copy of the weights instead of overwriting
the parameters on every forward; sampling normalises each row by its own sum (the
reference divides by the sum over the whole batch, models.py:194) and accepts a
torch.Generator or explicit uniforms.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

import dvq_hip
from dvq_hip import ops
from dvq_hip._lib import DVQ_BF16, DVQ_F32
from dvq_hip.rng import sample_rows


def weights_init(m):
    classname = m.__class__.__name__
    if classname.find('Conv') != -1 and hasattr(m, 'weight') and isinstance(getattr(m, 'weight'), torch.Tensor):
        nn.init.xavier_uniform_(m.weight.data)
        m.bias.data.fill_(0)


class GatedActivation(nn.Module):
    def forward(self, x):
        a, b = x.chunk(2, dim=1)
        return torch.tanh(a) * torch.sigmoid(b)


class GatedMaskedConv2d(nn.Module):
    def __init__(self, mask_type, dim, kernel, residual=True, n_classes=128):
        super().__init__()
        assert kernel % 2 == 1, "Kernel size must be odd"
        self.mask_type = mask_type
        self.residual = residual
        self.kernel = kernel
        self.class_cond_embedding = nn.Embedding(n_classes, 2 * dim)
        self.vert_stack = nn.Conv2d(dim, dim * 2, (kernel // 2 + 1, kernel), 1, (kernel // 2, kernel // 2))
        self.vert_to_horiz = nn.Conv2d(2 * dim, 2 * dim, 1)
        self.horiz_stack = nn.Conv2d(dim, dim * 2, (1, kernel // 2 + 1), 1, (0, kernel // 2))
        self.horiz_resid = nn.Conv2d(dim, dim, 1)
        self.gate = GatedActivation()

    def make_causal(self):
        self.vert_stack.weight.data[:, :, -1].zero_()
        self.horiz_stack.weight.data[:, :, :, -1].zero_()

    def forward(self, x_v, x_h, h):
        wv, wh = self.vert_stack.weight, self.horiz_stack.weight
        if self.mask_type == 'A':
            mv = torch.ones_like(wv); mv[:, :, -1] = 0
            mh = torch.ones_like(wh); mh[:, :, :, -1] = 0
            wv, wh = wv * mv, wh * mh
        k = self.kernel
        h = self.class_cond_embedding(h)
        h_vert = F.conv2d(x_v, wv, self.vert_stack.bias, 1, (k // 2, k // 2))[:, :, :x_v.size(-1), :]
        out_v = self.gate(h_vert + h[:, :, None, None])
        h_horiz = F.conv2d(x_h, wh, self.horiz_stack.bias, 1, (0, k // 2))[:, :, :, :x_h.size(-2)]
        v2h = self.vert_to_horiz(h_vert)
        out = self.gate(v2h + h_horiz + h[:, :, None, None])
        out_h = self.horiz_resid(out) + x_h if self.residual else self.horiz_resid(out)
        return out_v, out_h


class _PackedTaps:
    """Per-layer weights re-laid for the incremental evaluation: one matrix per kernel tap that can touch the
    grid, packed for the HIP products (dvq_prior_gemm) in A-operand fragment order -- bf16, or fp32 for the
    parity configuration; biases and class embeddings stay fp32."""

    def __init__(self, layer: GatedMaskedConv2d, code: int):
        k = layer.kernel
        pack = lambda w: ops.prior_pack_weight(w, code)
        wv = layer.vert_stack.weight.detach().float()
        wh = layer.horiz_stack.weight.detach().float()
        rows = k // 2 + 1
        live = range(rows - 1) if layer.mask_type == 'A' else range(rows)
        self.vert = {(a - k // 2, b - k // 2): pack(wv[:, :, a, b]) for a in live for b in range(k)}
        self.horiz = {b - k // 2: pack(wh[:, :, 0, b]) for b in live}
        self.v2h = pack(layer.vert_to_horiz.weight.detach().float()[:, :, 0, 0])
        self.resid = pack(layer.horiz_resid.weight.detach().float()[:, :, 0, 0])
        self.vert_bias = layer.vert_stack.bias.detach().float().contiguous()
        # the horizontal stack adds the projected vertical features before its gate (models.py:79-81)
        self.horiz_bias = (layer.horiz_stack.bias.detach().float() + layer.vert_to_horiz.bias.detach().float()).contiguous()
        self.resid_bias = layer.horiz_resid.bias.detach().float().contiguous()
        self.cond = layer.class_cond_embedding.weight.detach().float().contiguous()
        self.residual = layer.residual
        self.kernel = k
        self.mask_a = layer.mask_type == 'A'


class GatedPixelCNN(dvq_hip.DropsPackedCopies, nn.Module):
    _packed_defaults = {"_packed_key": lambda: None, "_packed": lambda: None, "_bufs": dict, "_cls": lambda: None}

    def __init__(self, input_dim=256, dim=128, n_layers=15, n_classes=128):
        super().__init__()
        self.dim = dim
        self.embedding = nn.Embedding(input_dim, dim)
        self.layers = nn.ModuleList()
        for i in range(n_layers):
            self.layers.append(GatedMaskedConv2d('A' if i == 0 else 'B', dim, 5 if i == 0 else 3, i != 0, n_classes))
        self.output_conv = nn.Sequential(nn.Conv2d(dim, 2048, 1), nn.ReLU(True), nn.Conv2d(2048, input_dim, 1))
        self.apply(weights_init)
        self.compute_dtype = torch.float32     # operand precision of the matrix products in generate()
        self._packed_key = None
        self._packed = None
        self._bufs = {}
        self.class_row1 = True                 # grid row 1: the taps on grid row 0 per class as well
        self._cls = None                       # per-class constants of the weights (see _class_rows)
        self.use_graph = True                  # replay generate() as a captured HIP graph
        self.split = 1                         # sub-batches evaluated side by side (own streams and graphs)
        self.split_min_rows = 2048             # ... for batches of at least this many rows
        self.vrow_min_rows = 3072              # fused vertical-row kernel (bf16) from this many rows per (sub-)batch
        self.class_rows = True                 # grid row 0's vertical stack and cell (0,0) once per class, not per row
        self.class_rows_min_rows = 1           # ... for batches of at least this many rows
        # Rows of a batch evaluated in class order: a tile's epilogue then reads few rows of the per-class tables instead
        # of one per sample (4096 grids of 128 classes drawn at random: 5.72 -> 5.50 ms; of one class: 5.27 -> 5.32 ms,
        # the price of the sort and the two permutations).  Same grids row for row.
        self.sort_by_class = True
        self.sort_min_rows = 2048              # ... for batches of at least this many rows

    def forward(self, x, label):
        """Logits [B, input_dim, H, W] of the grid x (int64 [B,H,W]) under class `label` (models.py:161-174).  On a HIP
        device in eval mode the masked convolutions are evaluated cell by cell on the dvq_prior kernels with x as the
        given history -- the same numbers (the convolutions are causal: cell (i,j) sees only earlier cells), no library
        convolution; in training mode or on CPU tensors the convolution form below runs (it carries autograd)."""
        if x.is_cuda and not self.training and not torch.is_grad_enabled():
            return self._forward_hip(x, label)
        return self.forward_convs(x, label)

    def forward_convs(self, x, label):
        """The reference's literal forward on framework convolutions: kept as the independent check of the incremental
        evaluation (generate_full) and for training."""
        shp = x.size() + (-1,)
        x = self.embedding(x.view(-1)).view(shp).permute(0, 3, 1, 2)
        x_v, x_h = x, x
        for layer in self.layers:
            x_v, x_h = layer(x_v, x_h, label)
        return self.output_conv(x_h)

    @torch.no_grad()
    def _forward_hip(self, x, label):
        B, H, W = x.shape
        dev = x.device
        self._prepared_hip()
        bufs = self._act_buffers(B, H, W, dev, part=("forward", 0))      # own buffers: a recorded generate() graph stays
        bufs["label"].copy_(label.reshape(-1).to(torch.int64))
        if self._by_class(B):
            self._class_rows((H, W), B >= self.vrow_min_rows)
        out = torch.empty(B, self.output_conv[2].out_channels, H, W, dtype=torch.float32, device=dev)
        self._run_hip(bufs, (H, W), B, None, False, None, forced=x.to(torch.int64).contiguous(), collect=out)
        return out

    # ------------------------------------------------------------------ sampling
    def _code(self):
        if self.compute_dtype == torch.bfloat16:
            return DVQ_BF16
        if self.compute_dtype == torch.float32:
            return DVQ_F32
        raise RuntimeError(f"GatedPixelCNN: compute_dtype {self.compute_dtype} (float32 or bfloat16 expected)")

    def _weights_key(self):
        return ((dvq_hip.weights_epoch(),) + tuple((p.data_ptr(), p._version) for p in self.parameters()), self._code())

    def _prepared_hip(self):
        key = self._weights_key()
        if key != self._packed_key:
            code = key[1]
            head = (ops.prior_pack_weight(self.output_conv[0].weight.detach().float()[:, :, 0, 0], code),
                    self.output_conv[0].bias.detach().float().contiguous(),
                    ops.prior_pack_weight(self.output_conv[2].weight.detach().float()[:, :, 0, 0], code),
                    self.output_conv[2].bias.detach().float().contiguous())
            self._packed = ([_PackedTaps(layer, code) for layer in self.layers], head)
            self._packed_key = key
            self._bufs = {}
            self._cls = None
        return self._packed

    def graph_ready(self, batch_size, shape=(3, 3)):
        """False when the next generate(..., uniforms=u) of this batch size will record a HIP graph (first call of a
        batch size, or weights or precision changed since): callers with asynchronous work of their own in flight (a
        pending collective) complete it first.  True when a recorded graph will be replayed or none is used."""
        if not self.use_graph or not self.embedding.weight.is_cuda:
            return True
        b = self._bufs.get((batch_size, shape[0], shape[1], str(self.embedding.weight.device)))
        return (self._weights_key() == self._packed_key and b is not None and b.get("graph") is not None
                and b.get("split") == self._plan(batch_size))

    def _act_buffers(self, B, H, W, dev, part=None):
        """Packed activation buffers for a batch of B grids, allocated once per (B, H, W); `part` tells the
        buffer sets of the sub-batches of one split batch apart.  A new batch size drops the old buffers."""
        key = (B, H, W, str(dev)) if part is None else (B, H, W, str(dev), part)
        if key not in self._bufs:
            if part is None:
                self._bufs = {}
            self._bufs[key] = self._new_buffers(B, H, W, dev)
        return self._bufs[key]

    def _new_buffers(self, B, H, W, dev):
        L, d = len(self.layers), self.dim
        code = self._code()
        e = lambda f, zero=False: ops.prior_packed_empty(B, f, dev, zero, code)
        return dict(
            XV=[[[e(d) for _ in range(W)] for _ in range(H)] for _ in range(L + 1)],
            HV=[[e(2 * d) for _ in range(W)] for _ in range(L)],
            XH=[[e(d) for _ in range(L + 1)] for _ in range(2)],
            OUT=e(d), HID=e(self.output_conv[0].out_channels), Z=e(d, True),
            logits=torch.empty(B, self.output_conv[2].out_channels, dtype=torch.float32, device=dev),
            emb=self.embedding.weight.detach().float().contiguous(),
            label=torch.zeros(B, dtype=torch.int64, device=dev),
            u=torch.zeros(B, H * W, dtype=torch.float64, device=dev),
            perm=torch.zeros(B, dtype=torch.int32, device=dev), graph=None, x=None)

    @torch.no_grad()
    def generate(self, x_start, label, shape=(3, 3), batch_size=64, generator=None, return_probs=False,
                 uniforms=None, _sorted=False):
        """Sample a grid of indices, cell by cell in raster order (models.py:176-198).
        `x_start` is accepted and ignored, as in the reference.  Cell (i,j) is drawn from
        softmax(logits[:, :, i, j]) with torch.multinomial (optionally `generator`) or, when
        `uniforms` [B, H*W] is given, by inverse CDF.

        The products run on the HIP kernels of dvq_prior_gemm in `compute_dtype`: float32 (exact fp32, the
        reference's arithmetic) or bfloat16 operands with fp32 accumulation.  With explicit uniforms the whole
        procedure is a fixed sequence of ~320 launches over persistent buffers; it is captured once per batch
        size into a HIP graph and replayed, because issuing that many small dependent launches from Python
        costs more than running them."""
        H, W = shape
        B = batch_size
        dev = self.embedding.weight.device
        if not self.embedding.weight.is_cuda:
            raise RuntimeError("GatedPixelCNN.generate needs the module on a HIP device (there is no CPU path)")
        if B == 0:                                 # an empty batch (a rank without objects): nothing to draw
            n_out = self.output_conv[2].out_channels
            empty = torch.zeros((0, H, W), dtype=torch.int64, device=dev)
            return (empty, torch.zeros((0, H * W, n_out), dtype=torch.float32, device=dev)) if return_probs else empty
        self._prepared_hip()
        bufs = self._act_buffers(B, H, W, dev)
        n_classes = self.layers[0].class_cond_embedding.num_embeddings
        ordered = (self.sort_by_class and B >= self.sort_min_rows and uniforms is not None and not return_probs
                   and not _sorted and n_classes <= ops.PRIOR_ORDER_MAX_CLASSES)
        if ordered:
            # Rows are independent and a row's arithmetic does not depend on its position, so the batch is evaluated in
            # class order (a tile's epilogue then reads a few rows of the per-class tables instead of one per sample)
            # and the grids are handed back in the caller's order: same grids, row for row.  One launch writes the
            # sorted labels, the uniforms in that order and the permutation (a stable counting sort).
            ops.prior_class_order(label.reshape(-1).to(torch.int64), uniforms, n_classes, bufs["label"], bufs["u"], bufs["perm"])
            uniforms = bufs["u"]
        else:
            bufs["label"].copy_(label.reshape(-1).to(torch.int64))
        if self._by_class(B):
            self._class_rows(shape, B >= self.vrow_min_rows)        # before any graph is recorded
        def caller_order(x):
            return ops.unpermute_rows(x.view(B, H * W), bufs["perm"]).view(B, H, W) if ordered else x

        if uniforms is None or return_probs or not self.use_graph:
            out = self._run_hip(bufs, shape, B, generator, return_probs, uniforms)
            return caller_order(out) if not return_probs else out
        if not ordered:
            bufs["u"].copy_(uniforms)
        if bufs.get("graph") is None or bufs.get("split") != self._plan(B):
            n_parts = self._n_parts(B)
            parts = None
            if n_parts > 1:
                # Rows are independent: the sub-batches run the same chain on streams of their own, one's launch
                # boundaries under the other's work.
                per = (B + n_parts - 1) // n_parts // 32 * 32
                cuts = [min(B, k * per) for k in range(n_parts)] + [B]
                parts = [(a, b, self._act_buffers(b - a, H, W, dev, part=(B, k)), torch.cuda.Stream(device=dev))
                         for k, (a, b) in enumerate(zip(cuts[:-1], cuts[1:]))]
            if parts is None:
                # (thread-local capture mode: another thread of the process -- the watchdog of a process group polling
                # its events -- must not invalidate the recording)
                self._run_hip(bufs, shape, B, None, False, bufs["u"])    # warm-up outside capture
                torch.cuda.synchronize(dev)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    bufs["x"] = self._run_hip(bufs, shape, B, None, False, bufs["u"])
                graphs = [g]
            else:
                # one graph per sub-batch, replayed on streams of their own (the branches of ONE graph were
                # seen to run one after the other)
                bufs["x"] = torch.zeros((B, H, W), dtype=torch.int64, device=dev)
                graphs = []
                for a, b, pb, st in parts:
                    self._run_hip(pb, shape, b - a, None, False, bufs["u"][a:b], x_out=bufs["x"][a:b])
                    torch.cuda.synchronize(dev)
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        pb["label"].copy_(bufs["label"][a:b])
                        self._run_hip(pb, shape, b - a, None, False, bufs["u"][a:b], x_out=bufs["x"][a:b])
                    graphs.append(g)
            bufs["graph"], bufs["split"], bufs["parts"] = graphs, self._plan(B), parts
        if bufs["parts"] is None:
            bufs["graph"][0].replay()
        else:
            main = torch.cuda.current_stream(dev)
            for g, (_, _, _, st) in zip(bufs["graph"], bufs["parts"]):
                st.wait_stream(main)
                with torch.cuda.stream(st):
                    g.replay()
            for _, _, _, st in bufs["parts"]:
                main.wait_stream(st)
        return caller_order(bufs["x"]) if ordered else bufs["x"].clone()

    def _plan(self, B):
        """What a recorded graph of batch size B depends on besides the weights."""
        return (self._n_parts(B), self.vrow_min_rows, self.class_rows, self.class_rows_min_rows, self.class_row1)

    def _n_parts(self, B):
        return self.split if (self.split > 1 and B >= self.split_min_rows) else 1

    def _gemm(self, jobs, M, N, epi, relu=False):
        ops.prior_gemm(jobs, M, N, epi, relu, self.embedding.weight, self._packed_key[1])

    def _vertical_row(self, bufs, i, M, shape, label, use_vrow, above_by_class=None):
        """Vertical stack of grid row i for all its cells: XV[l+1][i][:] and the pre-gate features HV[l][:].
        above_by_class: per layer None or, per cell, a [classes, 2d] table holding the taps on the row above (grid row 1,
        whose upper row is grid row 0: per class); those taps are dropped and the table's row is added before the gate."""
        H, W = shape
        layers = self._packed[0]
        code = self._packed_key[1]
        XV, HV, Z = bufs["XV"], bufs["HV"], bufs["Z"]
        d = self.dim
        kb = d // 32
        for l, t in enumerate(layers):
            folded = above_by_class[l] if above_by_class is not None else None
            above = (0,) if folded is not None else (-1, 0)
            if code == DVQ_BF16 and W == 3 and t.kernel == 3 and not t.mask_a and use_vrow:
                # all three cells of the row in one launch: every input column and every tap is fetched once
                # (smaller batches are latency-bound: per-cell launches)
                rows = [([XV[l][i + da][c] for c in range(3)], [t.vert[(da, db)] for db in (-1, 0, 1)])
                        for da in above if i + da >= 0]
                ops.prior_vrow(rows, kb, t.vert_bias, t.cond, label, [XV[l + 1][i][j] for j in range(3)],
                               [HV[l][j] for j in range(3)], M, 2 * d, pre_add=folded)
                continue
            jobs = []
            for j in range(W):
                taps = [(XV[l][i + da][j + db], w, kb) for (da, db), w in t.vert.items()
                        if (folded is None or da == 0) and 0 <= i + da < H and 0 <= j + db < W and (l > 0 or i + da < i)]
                if not taps:                                     # nothing above this row: bias only
                    taps = [(Z, next(iter(t.vert.values())), kb)]
                jobs.append(ops.prior_job(taps, t.vert_bias, cond=t.cond, label=label, out_packed=XV[l + 1][i][j],
                                          out_pre=HV[l][j], pre_add=None if folded is None else folded[j]))
            self._gemm(jobs, M, 2 * d, ops.PRIOR_GATE)

    def _horizontal_cell(self, bufs, i, j, M, label, cond_tables=None, v2h_tap=True, prev_tap=True):
        """Horizontal stack and output head of cell (i, j): two launches per layer (taps + projected vertical
        features + gate; residual 1x1), two for the head; leaves the cell's logits in bufs["logits"].
        cond_tables: per layer a [classes, 2d] table used instead of the class embedding -- it already holds class
        embedding + projected vertical features (grid row 0, where the vertical features depend on the class alone);
        the projection tap is then dropped (v2h_tap=False).  prev_tap=False: the table also holds the taps on the
        cell to the left from layer 1 on (cell (0, 1), whose left neighbour (0, 0) is per class), which are dropped."""
        layers, (wo1, bo1, wo2, bo2) = self._packed
        XV, HV, XHb, OUT, HID, logits = (bufs[k] for k in ("XV", "HV", "XH", "OUT", "HID", "logits"))
        L, d = len(layers), self.dim
        kb = d // 32
        XH, XHp = XHb[j & 1], XHb[(j & 1) ^ 1]
        for l, t in enumerate(layers):
            taps = []
            for db, w in t.horiz.items():
                c = j + db
                if c < 0:
                    continue
                if l == 0:
                    taps.append((XV[0][i][c], w, kb))
                elif c == j:
                    taps.append((XH[l], w, kb))
                elif c == j - 1 and prev_tap:
                    taps.append((XHp[l], w, kb))
            if v2h_tap:
                taps.append((HV[l][j], t.v2h, 2 * kb))
            table = cond_tables[l] if cond_tables is not None and cond_tables[l] is not None else t.cond
            self._gemm([ops.prior_job(taps, t.horiz_bias, cond=table, label=label, out_packed=OUT)], M, 2 * d, ops.PRIOR_GATE)
            self._gemm([ops.prior_job([(OUT, t.resid, kb)], t.resid_bias, resid=XH[l] if t.residual else None,
                                      out_packed=XH[l + 1])], M, d, ops.PRIOR_PACK)
        self._gemm([ops.prior_job([(XH[L], wo1, kb)], bo1, out_packed=HID)], M, bo1.numel(), ops.PRIOR_PACK, True)
        self._gemm([ops.prior_job([(HID, wo2, bo1.numel() // 32)], bo2, out_rows=logits)], M, logits.shape[1], ops.PRIOR_ROWS)

    # What the loop computes before anything has been drawn depends on a row's class alone: the vertical stack of
    # grid row 0 (nothing above it) and the whole of cell (0, 0) (nothing left of it), logits included.  Like the
    # packed weights they are constants of the checkpoint: evaluated for ONE row per class with the same kernels when
    # the weights are first used (_class_rows) and handed out to the rows of a batch by their label
    # (_first_row_by_class) -- equal to evaluating every row (to the bit when the batch takes the same product
    # kernel as the 128 class rows, to fp32 rounding otherwise).  The rest of grid row 0 reads the row's vertical
    # features only through their 1x1 projection into the gate: that projection, per class as well, is added to the
    # class-embedding table those cells' gate epilogue looks up, so the tap and the copies of the features go; so do
    # cell (0, 1)'s taps on cell (0, 0) and, in grid row 1, the vertical taps on grid row 0 (added before the gate by
    # label): what the batch is handed by label in the end is cell (0, 0)'s logits and nothing else.
    def _by_class(self, B):
        return self.class_rows and B >= self.class_rows_min_rows

    def _class_rows(self, shape, use_vrow):
        """Buffers of the per-class evaluation for this grid shape (and vertical-stack kernel), computed on first
        use after the weights changed."""
        dev = self.embedding.weight.device
        key = (shape, use_vrow, str(dev))
        if self._cls is None:
            self._cls = {}
        if key not in self._cls:
            V = self._packed[0][0].cond.shape[0]
            cls = self._new_buffers(V, shape[0], shape[1], dev)
            cls["label"].copy_(torch.arange(V, dtype=torch.int64, device=dev))
            self._vertical_row(cls, 0, V, shape, cls["label"], use_vrow)
            self._horizontal_cell(cls, 0, 0, V, cls["label"])
            # cells (0, 1..): their projected vertical features are per class too -- folded, with the class embedding,
            # into the table the gate epilogue looks up by label (one tap less in those cells' gate products)
            layers = self._packed[0]
            d = self.dim
            zero = torch.zeros(2 * d, dtype=torch.float32, device=dev)
            cls["cond_v2h"] = {}
            for j in range(1, shape[1]):
                tabs = []
                for l, t in enumerate(layers):
                    proj = torch.empty(V, 2 * d, dtype=torch.float32, device=dev)
                    taps = [(cls["HV"][l][j], t.v2h, 2 * d // 32)]
                    if j == 1 and l > 0 and -1 in t.horiz:
                        taps.append((cls["XH"][0][l], t.horiz[-1], d // 32))     # cell (0, 0)'s features, per class too
                    self._gemm([ops.prior_job(taps, zero, out_rows=proj)], V, 2 * d, ops.PRIOR_ROWS)
                    tabs.append((t.cond + proj).contiguous())
                cls["cond_v2h"][j] = tabs
            # grid row 1: from layer 1 on, its vertical stack reads grid row 0 (the taps one row up) only through
            # activations that are per class -- evaluated per class and added before the gate by label
            cls["above_row1"] = None
            if shape[0] > 1:
                rows1 = []
                for l, t in enumerate(layers):
                    if l == 0 or t.mask_a or not all((-1, db) in t.vert for db in (-1, 0, 1)):
                        rows1.append(None)
                        continue
                    tabs = []
                    for j in range(shape[1]):
                        taps = [(cls["XV"][l][0][j + db], t.vert[(-1, db)], d // 32) for db in (-1, 0, 1) if 0 <= j + db < shape[1]]
                        proj = torch.empty(V, 2 * d, dtype=torch.float32, device=dev)
                        self._gemm([ops.prior_job(taps, zero, out_rows=proj)], V, 2 * d, ops.PRIOR_ROWS)
                        tabs.append(proj)
                    rows1.append(tabs)
                cls["above_row1"] = rows1
            self._cls[key] = cls
        return self._cls[key]

    def _first_row_by_class(self, bufs, B, shape, label, use_vrow):
        H, W = shape
        layers = self._packed[0]
        L, d = len(layers), self.dim
        cls = self._class_rows(shape, use_vrow)
        V = cls["label"].shape[0]
        rows1 = cls["above_row1"]
        pairs = [(cls["XV"][l][0][j], bufs["XV"][l][0][j], d) for l in range(1, L) for j in range(W)
                 if H > 1 and (rows1 is None or rows1[l] is None or not self.class_row1)]   # read by row 1 where its taps are not folded
        if pairs:
            ops.prior_gather_rows(pairs, label, B, V, self._packed_key[1])
        ops.vq_gather(label, cls["logits"], out=bufs["logits"])

    def _run_hip(self, bufs, shape, B, generator, return_probs, uniforms, x_out=None, forced=None, collect=None):
        """Grid row by grid row: the vertical stack of the row, then per cell the horizontal stack, the head and
        the draw (with the embedding of the drawn code as the next cell's input).  With `forced` (int64 [B,H,W]) nothing
        is drawn: the given codes are the history, and the cells' logits are written to collect[:, :, i, j]."""
        H, W = shape
        code = self._packed_key[1]
        dev = self.embedding.weight.device
        XV, logits = bufs["XV"], bufs["logits"]
        emb = bufs["emb"]
        label = bufs["label"]
        x = torch.empty((B, H, W), dtype=torch.int64, device=dev) if x_out is None else x_out    # every cell is written below
        probs_all = []
        use_vrow = B >= self.vrow_min_rows
        by_class = self._by_class(B)
        for i in range(H):
            if i == 0 and by_class:
                self._first_row_by_class(bufs, B, shape, label, use_vrow)
            else:
                above = self._class_rows(shape, use_vrow)["above_row1"] if (i == 1 and by_class and self.class_row1) else None
                self._vertical_row(bufs, i, B, shape, label, use_vrow, above)
            for j in range(W):
                if not (i == 0 and j == 0 and by_class):
                    cls = self._class_rows(shape, use_vrow) if by_class else None
                    if by_class and i == 0:
                        self._horizontal_cell(bufs, i, j, B, label, cls["cond_v2h"][j], v2h_tap=False, prev_tap=j != 1)
                    else:
                        self._horizontal_cell(bufs, i, j, B, label)
                if forced is not None:
                    if collect is not None:
                        collect[:, :, i, j] = logits
                    x[:, i, j] = forced[:, i, j]
                    ops.prior_embed(x[:, i, j], emb, XV[0][i][j], code)
                    continue
                if uniforms is not None and not return_probs:
                    # softmax, draw, grid write and embedding of the drawn code in one launch
                    ops.prior_sample(logits, uniforms[:, i * W + j], x[:, i, j], emb, XV[0][i][j], code)
                    continue
                probs = F.softmax(logits, -1)
                probs = probs / probs.sum(dim=1, keepdim=True)
                if return_probs:
                    probs_all.append(probs)
                if uniforms is not None:
                    x[:, i, j] = sample_rows(probs, uniforms[:, i * W + j].contiguous())
                else:
                    x[:, i, j] = probs.multinomial(1, generator=generator).squeeze(1)
                ops.prior_embed(x[:, i, j], emb, XV[0][i][j], code)
        if return_probs:
            return x, torch.stack(probs_all, 1)
        return x

    @torch.no_grad()
    def generate_full(self, label, shape=(3, 3), batch_size=64, forced=None):
        """The reference's literal procedure (one full forward per cell), with the samples
        replaced by `forced` indices: returns the per-cell distributions [B, H*W, classes].
        Used to check generate() against."""
        x = torch.zeros((batch_size, *shape), dtype=torch.int64, device=self.embedding.weight.device)
        out = []
        for i in range(shape[0]):
            for j in range(shape[1]):
                probs = F.softmax(self.forward_convs(x, label)[:, :, i, j], -1)
                out.append(probs / probs.sum(dim=1, keepdim=True))
                x[:, i, j] = forced[:, i, j]
        return torch.stack(out, 1)
