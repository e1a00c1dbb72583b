"""Device layout of the dense (non-table) variables.

The kernels want the reference's variables packed differently from how TF 1.14
creates them:

* (T-SeqRec cell, time_aware_rnn.py:79-123) the input kernels of its two time gates join ``gru/wx`` as
  columns 3D..5D and their biases join ``gru/bx``; the two time kernels form ``gru/tsr_wt`` [2,D,D], the
  four time-input vectors ``gru/tsr_tvec`` [4,D];
* the GRU ``gates/kernel`` [2D,2D] and ``candidate/kernel`` [2D,D]
  (Model/Modules/time_aware_rnn.py:166-185) are split into their input rows,
  packed side by side as ``gru/wx`` [D,3D] (one hoisted GEMM), and their
  recurrent rows ``gru/wh_g`` [D,2D], ``gru/wh_c`` [D,D];
* the K and V projections of every decoder block
  (Model/Modules/time_aware_attention.py:251-253) are packed into one
  ``kv/w`` [D, 2*NB*D] because the keys are the same for every block;
* per block, ``dense/kernel`` and ``_time_input_w`` (:249,:269-271) form
  ``blk{i}/wqt`` [D,2D]; the five [1,L] time-gate rows (:295-312) form
  ``blk{i}/tparams`` [5,L].

All segments live in ONE flat float32 buffer (parameters, gradients, Adam m and
v share the layout) so that clipping and Adam are single launches.  Segment
offsets are multiples of 4 floats (16 B).  ``pack``/``unpack`` convert to and
from TF-named arrays for weight injection and checkpoint interchange.
"""
import collections

import numpy as np

from .variables import GRU_DEAD, GRU_USED, MTAM_VARIANTS, SHORT_LN, TIME_GATE, TSR_VEC, gru_scope, head_ln_scope

Segment = collections.namedtuple("Segment", "name offset shape size")


def _align4(n):
    return (n + 3) // 4 * 4


class DenseLayout(object):

    def __init__(self, model, D, L, num_blocks):
        self.model, self.D, self.L, self.NB = model, D, L, num_blocks
        self.mtam = model in MTAM_VARIANTS
        self.cfg = MTAM_VARIANTS.get(model)
        segs = [("dense4emb/w", (2 * D, D))]
        if self.mtam:
            xw = 5 if self.cfg["gru"] == "seqrec" else 3
            segs += [("gru/wx", (D, xw * D)), ("gru/bx", (xw * D,)), ("gru/wh_g", (D, 2 * D)),
                     ("gru/wh_c", (D, D))]
            if self.cfg["gru"] == "time":
                segs.append(("gru/tvec", (8, D)))
            if self.cfg["gru"] == "seqrec":
                segs += [("gru/tsr_wt", (2, D, D)), ("gru/tsr_tvec", (4, D))]
            if self.cfg["short_ln"]:
                segs.append(("short/ln", (2, D)))
            if self.cfg["attention"]:
                segs += [("kv/w", (D, 2 * num_blocks * D)), ("kv/b", (2 * num_blocks * D,))]
                for i in range(num_blocks):
                    segs += [("blk%d/wqt" % i, (D, 2 * D)), ("blk%d/bq" % i, (D,)),
                             ("blk%d/tparams" % i, (5, L)), ("blk%d/ln" % i, (2, D))]
        else:
            for i in range(num_blocks):
                segs += [("blk%d/wqkv" % i, (D, 3 * D)), ("blk%d/bqkv" % i, (3 * D,)),
                         ("blk%d/wt" % i, (D, D)), ("blk%d/tparams" % i, (5, L, L)),
                         ("blk%d/ln" % i, (2, D))]
        segs.append(("head/ln", (2, D)))
        if self.mtam and self.cfg.get("head") == "concat":
            segs.append(("head/output_w", (2 * D, D)))
        self.segments = collections.OrderedDict()
        off = 0
        for name, shape in segs:
            size = int(np.prod(shape))
            self.segments[name] = Segment(name, off, shape, size)
            off = _align4(off + size)
        self.total = off

    def view(self, flat, name):
        s = self.segments[name]
        return flat[s.offset:s.offset + s.size].view(*s.shape) if hasattr(flat, "view") and not isinstance(flat, np.ndarray) \
            else flat[s.offset:s.offset + s.size].reshape(s.shape)

    # ------------------------------------------------------------ TF <-> native
    def _scopes(self):
        if self.mtam:
            blocks = ["NextItemDecoder/decoder/num_blocks_%d/" % i for i in range(self.NB)] \
                if self.cfg["attention"] else []
            return blocks, "vanilla_attention", head_ln_scope(self.model)
        return ["UserHistoryEncoder/encoder/num_blocks_%d/" % i for i in range(self.NB)], "self_attention", \
            "UserHistoryEncoder/LayerNorm/"

    def pack(self, tf_vars):
        """TF-named arrays -> flat float32 numpy buffer (pad floats are zero)."""
        D = self.D
        flat = np.zeros(self.total, np.float32)

        def put(name, arr):
            s = self.segments[name]
            flat[s.offset:s.offset + s.size] = np.asarray(arr, np.float32).reshape(-1)

        put("dense4emb/w", tf_vars["position_embedding/dense4emb/kernel"])
        scopes, inner, head = self._scopes()
        if self.mtam:
            GRU_SCOPE = gru_scope(self.model)
            Wg, Wc = tf_vars[GRU_SCOPE + "gates/kernel"], tf_vars[GRU_SCOPE + "candidate/kernel"]
            wx = [Wg[:D], Wc[:D]]
            bx = [tf_vars[GRU_SCOPE + "gates/bias"], tf_vars[GRU_SCOPE + "candidate/bias"]]
            if self.cfg["gru"] == "seqrec":
                wx += [tf_vars[GRU_SCOPE + "_time_kernel_w1"], tf_vars[GRU_SCOPE + "_time_kernel_w2"]]
                bx += [tf_vars[GRU_SCOPE + "_time_bias1"], tf_vars[GRU_SCOPE + "_time_bias2"]]
                put("gru/tsr_wt", np.stack([tf_vars[GRU_SCOPE + "_time_kernel_t1"], tf_vars[GRU_SCOPE + "_time_kernel_t2"]]))
                put("gru/tsr_tvec", np.stack([tf_vars[GRU_SCOPE + n] for n in TSR_VEC]))
            put("gru/wx", np.concatenate(wx, axis=1))
            put("gru/bx", np.concatenate(bx))
            put("gru/wh_g", Wg[D:])
            put("gru/wh_c", Wc[D:])
            if self.cfg["gru"] == "time":
                put("gru/tvec", np.stack([tf_vars[GRU_SCOPE + n] for n in GRU_USED]))
            if self.cfg["short_ln"]:
                put("short/ln", np.stack([tf_vars[SHORT_LN + "beta"], tf_vars[SHORT_LN + "gamma"]]))
            if scopes:
                put("kv/w", np.concatenate([np.concatenate([tf_vars[s + "dense_1/kernel"], tf_vars[s + "dense_2/kernel"]],
                                                           axis=1) for s in scopes], axis=1))
                put("kv/b", np.concatenate([np.concatenate([tf_vars[s + "dense_1/bias"], tf_vars[s + "dense_2/bias"]])
                                            for s in scopes]))
            for i, s in enumerate(scopes):
                a = s + inner + "/"
                put("blk%d/wqt" % i, np.concatenate([tf_vars[s + "dense/kernel"], tf_vars[a + "_time_input_w"]], axis=1))
                put("blk%d/bq" % i, tf_vars[s + "dense/bias"])
                put("blk%d/tparams" % i, np.concatenate([tf_vars[a + n] for n in TIME_GATE], axis=0))
                put("blk%d/ln" % i, np.stack([tf_vars[a + "ln/Variable"], tf_vars[a + "ln/Variable_1"]]))
        else:
            for i, s in enumerate(scopes):
                a = s + inner + "/"
                put("blk%d/wqkv" % i, np.concatenate([tf_vars[s + "dense/kernel"], tf_vars[s + "dense_1/kernel"],
                                                       tf_vars[s + "dense_2/kernel"]], axis=1))
                put("blk%d/bqkv" % i, np.concatenate([tf_vars[s + "dense/bias"], tf_vars[s + "dense_1/bias"],
                                                       tf_vars[s + "dense_2/bias"]]))
                put("blk%d/wt" % i, tf_vars[a + "_time_input_w"])
                put("blk%d/tparams" % i, np.stack([tf_vars[a + n] for n in TIME_GATE]))
                put("blk%d/ln" % i, np.stack([tf_vars[a + "ln/Variable"], tf_vars[a + "ln/Variable_1"]]))
        put("head/ln", np.stack([tf_vars[head + "beta"], tf_vars[head + "gamma"]]))
        if "head/output_w" in self.segments:
            put("head/output_w", tf_vars["output_w"])
        return flat

    def unpack(self, flat):
        """flat buffer (numpy) -> TF-named arrays (live variables only)."""
        D, NB = self.D, self.NB
        get = lambda name: np.array(self.view(flat, name))
        out = collections.OrderedDict()
        out["position_embedding/dense4emb/kernel"] = get("dense4emb/w")
        scopes, inner, head = self._scopes()
        if self.mtam:
            GRU_SCOPE = gru_scope(self.model)
            wx, bx = get("gru/wx"), get("gru/bx")
            out[GRU_SCOPE + "gates/kernel"] = np.concatenate([wx[:, :2 * D], get("gru/wh_g")], axis=0)
            out[GRU_SCOPE + "gates/bias"] = bx[:2 * D]
            out[GRU_SCOPE + "candidate/kernel"] = np.concatenate([wx[:, 2 * D:3 * D], get("gru/wh_c")], axis=0)
            out[GRU_SCOPE + "candidate/bias"] = bx[2 * D:3 * D]
            if self.cfg["gru"] == "seqrec":
                wt, tv4 = get("gru/tsr_wt"), get("gru/tsr_tvec")
                out[GRU_SCOPE + "_time_kernel_w1"], out[GRU_SCOPE + "_time_kernel_w2"] = wx[:, 3 * D:4 * D], wx[:, 4 * D:]
                out[GRU_SCOPE + "_time_bias1"], out[GRU_SCOPE + "_time_bias2"] = bx[3 * D:4 * D], bx[4 * D:]
                out[GRU_SCOPE + "_time_kernel_t1"], out[GRU_SCOPE + "_time_kernel_t2"] = wt[0], wt[1]
                for j, n in enumerate(TSR_VEC):
                    out[GRU_SCOPE + n] = tv4[j]
            if self.cfg["gru"] == "time":
                tv = get("gru/tvec")
                for j, n in enumerate(GRU_USED):
                    out[GRU_SCOPE + n] = tv[j]
            if self.cfg["short_ln"]:
                sl_ = get("short/ln")
                out[SHORT_LN + "beta"], out[SHORT_LN + "gamma"] = sl_[0], sl_[1]
            kvw, kvb = (get("kv/w"), get("kv/b")) if scopes else (None, None)
            for i, s in enumerate(scopes):
                a = s + inner + "/"
                wqt = get("blk%d/wqt" % i)
                out[s + "dense/kernel"], out[a + "_time_input_w"] = wqt[:, :D], wqt[:, D:]
                out[s + "dense/bias"] = get("blk%d/bq" % i)
                out[s + "dense_1/kernel"] = kvw[:, 2 * i * D:(2 * i + 1) * D]
                out[s + "dense_2/kernel"] = kvw[:, (2 * i + 1) * D:(2 * i + 2) * D]
                out[s + "dense_1/bias"] = kvb[2 * i * D:(2 * i + 1) * D]
                out[s + "dense_2/bias"] = kvb[(2 * i + 1) * D:(2 * i + 2) * D]
                tp = get("blk%d/tparams" % i)
                for j, n in enumerate(TIME_GATE):
                    out[a + n] = tp[j:j + 1]
                ln = get("blk%d/ln" % i)
                out[a + "ln/Variable"], out[a + "ln/Variable_1"] = ln[0], ln[1]
        else:
            for i, s in enumerate(scopes):
                a = s + inner + "/"
                w, bqkv = get("blk%d/wqkv" % i), get("blk%d/bqkv" % i)
                for j, layer in enumerate(("dense", "dense_1", "dense_2")):
                    out[s + layer + "/kernel"] = w[:, j * D:(j + 1) * D]
                    out[s + layer + "/bias"] = bqkv[j * D:(j + 1) * D]
                out[a + "_time_input_w"] = get("blk%d/wt" % i)
                tp = get("blk%d/tparams" % i)
                for j, n in enumerate(TIME_GATE):
                    out[a + n] = tp[j]
                ln = get("blk%d/ln" % i)
                out[a + "ln/Variable"], out[a + "ln/Variable_1"] = ln[0], ln[1]
        hl = get("head/ln")
        out[head + "beta"], out[head + "gamma"] = hl[0], hl[1]
        if "head/output_w" in self.segments:
            out["output_w"] = get("head/output_w")
        return out

    def dead_names(self):
        """Variables the reference creates but never updates (gradient None)."""
        scopes, inner, _ = self._scopes()
        names = [s + inner + "/time_output_w3" for s in scopes]
        if self.mtam and self.cfg["gru"] == "time":
            names += [gru_scope(self.model) + n for n in GRU_DEAD]
        return names
