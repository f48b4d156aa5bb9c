"""Host-side mirror of the reference model object.

`Clairvoyante` keeps the exact method / attribute surface of
/root/reference/clairvoyante/clairvoyante_v3.py:5-284 (identical in
clairvoyante_v3_slim.py) so that the reference's drivers (`callVar.py`, `train.py`,
`evaluate.py`) work unchanged against it; every method that was one
`tf.Session.run` is one call into the HIP library (include/clairvoyante_amd.h).
PyTorch is used for device memory, streams and `torch.distributed` only.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from . import param

PARAM_NAMES = [
    "conv1/kernel", "conv1/bias", "conv2/kernel", "conv2/bias", "conv3/kernel", "conv3/bias",
    "fc4/kernel", "fc4/bias", "fc5/kernel", "fc5/bias",
    "YBaseChangeSigmoid/kernel", "YBaseChangeSigmoid/bias",
    "YZygosityFC/kernel", "YZygosityFC/bias",
    "YVarTypeFC/kernel", "YVarTypeFC/bias",
    "YIndelLengthFC/kernel", "YIndelLengthFC/bias",
]


def _require_gpu():
    if not torch.cuda.is_available():
        raise _lib.CvError("clairvoyante_amd needs an AMD GPU (torch.cuda.is_available() is False); "
                           "there is no CPU fallback")


class Clairvoyante(object):
    """Drop-in for clairvoyante_v3.Clairvoyante (v3.py:7-29): same constructor
    keywords, same defaults; `pollSize*` = None / (1,1) means "no pooling layer"
    (the slim topology)."""

    def __init__(self, inputShape=(2 * param.flankingBaseNum + 1, 4, param.matrixNum),
                 outputShape1=(4,), outputShape2=(2,), outputShape3=(4,), outputShape4=(6,),
                 kernelSize1=(1, 4), kernelSize2=(2, 4), kernelSize3=(3, 4),
                 pollSize1=(5, 1), pollSize2=(4, 1), pollSize3=(3, 1),
                 numFeature1=16, numFeature2=32, numFeature3=48,
                 hiddenLayerUnits4=336, hiddenLayerUnits5=168,
                 initialLearningRate=param.initialLearningRate, learningRateDecay=param.learningRateDecay,
                 dropoutRateFC4=param.dropoutRateFC4, dropoutRateFC5=param.dropoutRateFC5,
                 l2RegularizationLambda=param.l2RegularizationLambda,
                 l2RegularizationLambdaDecay=param.l2RegularizationLambdaDecay,
                 device=None):
        if tuple(inputShape) != (33, 4, 4) or (tuple(outputShape1), tuple(outputShape2),
                                               tuple(outputShape3), tuple(outputShape4)) != ((4,), (2,), (4,), (6,)):
            raise ValueError("input [33,4,4] and heads 4/2/4/6 are fixed by the tensor format")
        for ks in (kernelSize1, kernelSize2, kernelSize3):
            if ks[1] != 4:
                raise ValueError("kernel width must be 4 (one tap per base)")
        if dropoutRateFC5 != 0.0:
            raise ValueError("dropoutRateFC5 other than 0.0 is not supported (reference default, param.py:25)")
        self.inputShape = inputShape
        self.outputShape1 = outputShape1; self.outputShape2 = outputShape2
        self.outputShape3 = outputShape3; self.outputShape4 = outputShape4
        self.kernelSize1 = kernelSize1; self.kernelSize2 = kernelSize2; self.kernelSize3 = kernelSize3
        self.pollSize1 = pollSize1; self.pollSize2 = pollSize2; self.pollSize3 = pollSize3
        self.numFeature1 = numFeature1; self.numFeature2 = numFeature2; self.numFeature3 = numFeature3
        self.hiddenLayerUnits4 = hiddenLayerUnits4; self.hiddenLayerUnits5 = hiddenLayerUnits5
        self.learningRateVal = initialLearningRate; self.learningRateDecay = learningRateDecay
        self.dropoutRateFC4Val = dropoutRateFC4; self.dropoutRateFC5Val = dropoutRateFC5
        self.l2RegularizationLambdaVal = l2RegularizationLambda
        self.l2RegularizationLambdaDecay = l2RegularizationLambdaDecay
        self.trainLossRTVal = None; self.trainSummaryRTVal = None; self.getLossLossRTVal = None
        self.predictBaseRTVal = None; self.predictZygosityRTVal = None
        self.predictVarTypeRTVal = None; self.predictIndelLengthRTVal = None

        _require_gpu()
        self._lib = _lib.load()
        if device is None:
            device = torch.cuda.current_device()
        self.device = torch.device("cuda", int(device))
        arch = _lib.CvArch()
        arch.kh[:] = [kernelSize1[0], kernelSize2[0], kernelSize3[0]]
        arch.cout[:] = [numFeature1, numFeature2, numFeature3]
        arch.pool[:] = [1 if p is None else p[0] for p in (pollSize1, pollSize2, pollSize3)]
        arch.fc4, arch.fc5 = hiddenLayerUnits4, hiddenLayerUnits5
        self._arch = arch
        h = ctypes.c_void_p()
        _lib.check(self._lib.cv_create(ctypes.byref(arch), self.device.index, ctypes.byref(h)))
        self._h = h
        self._shapes = {}
        for i in range(_lib.NUM_PARAMS):
            name = ctypes.c_char_p()
            nd = ctypes.c_int()
            dims = (ctypes.c_int64 * 4)()
            _lib.check(self._lib.cv_param_info(self._h, i, ctypes.byref(name), ctypes.byref(nd), dims))
            self._shapes[name.value.decode()] = tuple(int(d) for d in dims[:nd.value])
        self.numParameters = int(sum(int(np.prod(s)) for s in self._shapes.values()))
        self._adam_t = 0           # optimizer step count (beta*_power in the checkpoint)
        self._train_step = 0       # counter for the dropout stream
        self._dropout_seed = int.from_bytes(os.urandom(8), "little")   # reference is unseeded (selu.py:55)
        self._seed_rng = np.random.RandomState()
        self._bucket = None        # gradient bucket (torch tensor), bound on the first training step
        self._carry = ([0.0] * 6, 0)   # losses of deferred steps already read from the device accumulator
        self._deferred = 0             # trainDeferred steps whose losses still sit in the device accumulator
        self._keep = None

    # ---- helpers --------------------------------------------------------------
    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    accepts_device_batches = True      # train/getLoss/predict take torch tensors that already live on self.device

    def _to_dev(self, a, last):
        if torch.is_tensor(a):
            t = a.to(device=self.device, dtype=torch.float32).contiguous()
        else:
            # (a pageable source: the runtime's own staged copy moves 138 MB in ~4 ms, ~34 GB/s, on this platform; copying
            # through page-locked pieces of our own was measured 7x slower -- profiles/r03/small_batch_staged.txt)
            t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.device)
        return t.reshape((-1,) + last)

    def setOption(self, key, value):
        _lib.check(self._lib.cv_set_option(self._h, key.encode(), int(value)))

    def paramShapes(self):
        return dict(self._shapes)

    def setParameter(self, name, value):
        v = np.ascontiguousarray(value, dtype=np.float32)
        if tuple(v.shape) != self._shapes[name]:
            raise ValueError("%s: shape %s, expected %s" % (name, v.shape, self._shapes[name]))
        _lib.check(self._lib.cv_set_param(self._h, name.encode(), v.ctypes.data_as(ctypes.c_void_p),
                                          v.size, self._stream()))

    def getParameter(self, name):
        v = np.empty(self._shapes[name], dtype=np.float32)
        _lib.check(self._lib.cv_get_param(self._h, name.encode(), v.ctypes.data_as(ctypes.c_void_p),
                                          v.size, self._stream()))
        return v

    def setParameters(self, params):
        for n in PARAM_NAMES:
            self.setParameter(n, params[n])

    def getParameters(self):
        return {n: self.getParameter(n) for n in PARAM_NAMES}

    # ---- reference surface ------------------------------------------------------
    def init(self):
        """tf.global_variables_initializer (v3.py:177): variance_scaling_initializer(
        factor=2.0, FAN_IN, uniform=False) = truncated normal, sigma sqrt(1.3*2/fan_in)
        cut at 2 sigma, for conv/fc4/fc5 (v3.py:57); glorot_uniform for the heads
        (tf.layers.dense default, v3.py:125-135); zero biases; zero Adam slots."""
        rng = self._seed_rng
        for name, shp in self._shapes.items():
            if name.endswith("bias"):
                v = np.zeros(shp, dtype=np.float32)
            else:
                fan_in = int(np.prod(shp[:-1])); fan_out = shp[-1]
                if name.startswith("Y"):
                    lim = np.sqrt(6.0 / (fan_in + fan_out))
                    v = rng.uniform(-lim, lim, shp).astype(np.float32)
                else:
                    std = np.sqrt(1.3 * 2.0 / fan_in)
                    t = rng.standard_normal(shp)
                    bad = np.abs(t) > 2.0
                    while bad.any():
                        t[bad] = rng.standard_normal(int(bad.sum()))
                        bad = np.abs(t) > 2.0
                    v = (t * std).astype(np.float32)
            self.setParameter(name, v)
        self._zero_adam()

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.cv_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def predict_device(self, x_dev, out16=None):
        """Fast path: x_dev [n,33,4,4] fp32 on this GPU -> out16 [n,16] device tensor
        (no host copies, asynchronous on the current stream)."""
        n = x_dev.shape[0]
        if out16 is None:
            out16 = torch.empty((n, _lib.NUM_OUT), dtype=torch.float32, device=self.device)
        _lib.check(self._lib.cv_forward(self._h, ctypes.c_void_p(x_dev.data_ptr()), n,
                                        ctypes.c_void_p(out16.data_ptr()), self._stream()))
        return out16

    # predict(numpy): batches of at least PIPE_MIN candidates go to the device in parts, part k + 1's host-to-device copy on
    # a copy stream under part k's kernels (round 5: one synchronous copy, the pass, one copy back -- nothing overlapped --
    # gave a host caller 8.7 M candidates/s of the device's 18.7).  The copy moves 26 candidates/us, the kernels take 19:
    # a part may be up to ~1.4 x the one before without the pass waiting for its input, and the first one is small (its copy
    # is the only one nothing hides): a tenth of the batch, then x 1.6, at most PIPE_PART_MAX.  The 16 outputs go back per
    # part on a third stream, enqueued AFTER the last input copy (a device-to-host copy in flight stalls the runtime's
    # staged pageable copy: 6.7 against 4.6 ms for 65 536 candidates, profiles/r06/host_copy_probe.txt).
    PIPE_MIN, PIPE_FIRST_MIN, PIPE_GROW, PIPE_PART_MAX = 24576, 4096, 1.6, 65536

    @classmethod
    def _pipe_cuts(cls, n):
        """[0, c1, c2, ..., n]: part boundaries (multiples of 16) of a host batch of n candidates"""
        cuts, size = [0], max(cls.PIPE_FIRST_MIN, n // 10)
        while cuts[-1] < n:
            step = min(int(size), cls.PIPE_PART_MAX) // 16 * 16
            nxt = cuts[-1] + step
            if n - nxt < step // 2:          # a short tail joins the last part
                nxt = n
            cuts.append(min(nxt, n))
            size *= cls.PIPE_GROW
        return cuts

    def _host_streams(self):
        st = getattr(self, "_hstreams", None)
        if st is None:
            st = self._hstreams = (torch.cuda.Stream(device=self.device), torch.cuda.Stream(device=self.device))
        return st

    def _predict_host(self, XArray):
        with torch.cuda.device(self.device):
            if torch.is_tensor(XArray):
                out = self.predict_device(self._to_dev(XArray, (33, 4, 4))).cpu().numpy()
            else:
                x = np.ascontiguousarray(XArray, dtype=np.float32).reshape((-1, 33, 4, 4))
                n = x.shape[0]
                if n < self.PIPE_MIN:
                    out = self.predict_device(torch.from_numpy(x).to(self.device)).cpu().numpy()
                else:
                    out = self._predict_host_parts(x, n)
        # (views of ONE fresh [n,16] array: the four heads side by side, as the kernels store them)
        return out[:, 0:4], out[:, 4:6], out[:, 6:10], out[:, 10:16]

    def _predict_host_parts(self, x, n):
        cut = self._pipe_cuts(n)
        main = torch.cuda.current_stream(self.device)
        cs, ds = self._host_streams()
        dev = torch.empty((n, 33, 4, 4), dtype=torch.float32, device=self.device)
        out = torch.empty((n, _lib.NUM_OUT), dtype=torch.float32, device=self.device)
        host = torch.empty((n, _lib.NUM_OUT), dtype=torch.float32, pin_memory=True)
        cs.wait_stream(main)                  # the fresh buffers may still be in use by work queued on the caller's stream
        done = []
        for lo, hi in zip(cut[:-1], cut[1:]):
            with torch.cuda.stream(cs):
                dev[lo:hi].copy_(torch.from_numpy(x[lo:hi]))       # (host-synchronous: the runtime stages pageable memory)
                ev = torch.cuda.Event(); ev.record(cs)
            main.wait_event(ev)
            self.predict_device(dev[lo:hi], out[lo:hi])
            d = torch.cuda.Event(); d.record(main); done.append(d)
        with torch.cuda.stream(ds):
            for (lo, hi), d in zip(zip(cut[:-1], cut[1:]), done):
                ds.wait_event(d)
                host[lo:hi].copy_(out[lo:hi], non_blocking=True)
        ds.synchronize()                      # everything above is complete: the buffers may go back to their pools
        return host.numpy()

    def predict(self, XArray):
        """v3.py:257-267 -> (base [n,4], zygosity [n,2], varType [n,4], indelLength [n,6])"""
        return self._predict_host(XArray)

    def predictNoRT(self, XArray):
        """v3.py:269-280: results land in predict*RTVal (called from a worker thread)."""
        self.predictBaseRTVal = None; self.predictZygosityRTVal = None
        self.predictVarTypeRTVal = None; self.predictIndelLengthRTVal = None
        (self.predictBaseRTVal, self.predictZygosityRTVal,
         self.predictVarTypeRTVal, self.predictIndelLengthRTVal) = self._predict_host(XArray)

    def getActivation(self, layer, n):
        """Intermediate of the last pass in the reference's layout (debug / parity); layers 6 / 7 are the
        alpha-dropout keep mask (times its factor a) and output of the last train / getLoss slice."""
        a = self._arch
        hp = [33 - (a.pool[0] - 1)]
        hp.append(hp[0] - (a.pool[1] - 1)); hp.append(hp[1] - (a.pool[2] - 1))
        shp = {1: (n, hp[0], 4, a.cout[0]), 2: (n, hp[1], 4, a.cout[1]), 3: (n, hp[2], 4, a.cout[2]),
               4: (n, a.fc4), 5: (n, a.fc5),
               6: (n, a.fc4), 7: (n, a.fc4),            # last TRAINING slice: a*keep mask of fc4, dropout4 output
               11: (n, hp[0], 4, a.cout[0]), 12: (n, hp[1], 4, a.cout[1]), 13: (n, hp[2], 4, a.cout[2]),   # its pooled maps
               21: (n, 33, 4, a.cout[0]), 22: (n, hp[0], 4, a.cout[1]), 23: (n, hp[1], 4, a.cout[2])}[layer]   # its pre-activation gradients
        dst = torch.empty(shp, dtype=torch.float32, device=self.device)
        _lib.check(self._lib.cv_get_activation(self._h, layer, ctypes.c_void_p(dst.data_ptr()), n,
                                               self._stream()))
        return dst

    # ---- training ---------------------------------------------------------------
    def _zero_adam(self):
        """Adam slots m / v and the step count back to their initial state (what
        tf.global_variables_initializer does to "<var>/Adam", "<var>/Adam_1", beta*_power; v3.py:177)."""
        with torch.cuda.device(self.device):
            z = torch.zeros(self.numParameters, dtype=torch.float32, device=self.device)
            for which in (2, 3):
                _lib.check(self._lib.cv_flat_copy(self._h, which, ctypes.c_void_p(z.data_ptr()), 1, self._stream()))
            torch.cuda.current_stream(self.device).synchronize()
        self._adam_t = 0

    def _losses(self, fn, *args):
        losses = (ctypes.c_double * 6)()
        _lib.check(fn(*args, losses, self._stream()))
        return list(losses)

    def _loss_only(self, batchX, batchY):
        with torch.cuda.device(self.device):
            x = self._to_dev(batchX, (33, 4, 4)); y = self._to_dev(batchY, (16,))
            l = self._losses(self._lib.cv_loss, self._h, ctypes.c_void_p(x.data_ptr()),
                             ctypes.c_void_p(y.data_ptr()), x.shape[0])
        return np.float32(l[5])

    def _ensure_bucket(self):
        """The gradient bucket (loss header + flat gradient, include/clairvoyante_amd.h) lives in a torch tensor
        so that torch.distributed reduces it in place; the library writes into it (cv_bind_grad_bucket)."""
        if getattr(self, "_bucket", None) is None:
            cnt = ctypes.c_int64(); hdr = ctypes.c_int64(); dense = ctypes.c_int64()
            _lib.check(self._lib.cv_grad_bucket_info(self._h, ctypes.byref(cnt), ctypes.byref(hdr), ctypes.byref(dense)))
            self._bucket = torch.zeros(cnt.value, dtype=torch.float32, device=self.device)
            self._bucket_header, self._bucket_dense = int(hdr.value), int(dense.value)
            _lib.check(self._lib.cv_bind_grad_bucket(self._h, ctypes.c_void_p(self._bucket.data_ptr()), cnt.value))
        return self._bucket

    def gradients(self):
        """flat gradient of the last step (a view of the bucket behind its loss header)"""
        return self._ensure_bucket()[self._bucket_header:]

    def _enqueue_step(self, batchX, batchY):
        """One optimizer step enqueued on the current stream, no host synchronisation: forward + backward
        (cv_grad_async), gradient / loss exchange over the ranks (parallel.exchange_bucket: the dense 95 % of the
        bucket while the convolution backward pass still runs), Adam, losses added to the device accumulator."""
        from . import parallel
        x = self._to_dev(batchX, (33, 4, 4)); y = self._to_dev(batchY, (16,))
        self._train_step += 1
        self._ensure_bucket()
        comm = parallel.comm_stream(self)
        _lib.check(self._lib.cv_grad_async(self._h, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()),
                                           x.shape[0], ctypes.c_float(self.dropoutRateFC4Val),
                                           ctypes.c_float(self.l2RegularizationLambdaVal),
                                           ctypes.c_uint64(self._dropout_seed & 0xFFFFFFFFFFFFFFFF),
                                           ctypes.c_uint64(self._train_step), self._stream(),
                                           ctypes.c_void_p(comm.cuda_stream) if comm is not None else None))
        parallel.exchange_bucket(self, comm)
        self._adam_t += 1
        _lib.check(self._lib.cv_apply_adam_accumulate(self._h, ctypes.c_float(self.learningRateVal),
                                                      ctypes.c_float(self.l2RegularizationLambdaVal),
                                                      self._adam_t, self._stream()))      # Adam + losses to the accumulator
        self._keep = (x, y)          # the step is still running: the batch stays referenced until the next one

    def _read_acc(self):
        losses = (ctypes.c_double * 6)(); steps = ctypes.c_int64()
        _lib.check(self._lib.cv_loss_read(self._h, losses, ctypes.byref(steps), 1, self._stream()))
        return list(losses), int(steps.value)

    def readLosses(self, reset=True):
        """-> ([loss1..loss4, lossL2, total] summed over the trainDeferred steps since the last reset, number of
        steps); synchronises.  Global-batch values under data parallelism."""
        with torch.cuda.device(self.device):
            l, n = self._read_acc()
        l = [u + v for u, v in zip(l, self._carry[0])]; n += self._carry[1]
        self._carry = ([0.0] * 6, 0) if reset else (l, n)
        self._deferred = 0
        return l, n

    def trainDeferred(self, batchX, batchY):
        """trainNoRT without the host round trip: the step is enqueued and its losses are added to the device
        accumulator (readLosses); what train.run_epoch uses -- train.py:113-114 only needs the epoch's sum."""
        with torch.cuda.device(self.device):
            self._enqueue_step(batchX, batchY)
            self._deferred += 1

    def _train_step_impl(self, batchX, batchY):
        with torch.cuda.device(self.device):
            if self._deferred:
                self.readLosses(reset=False)      # park what trainDeferred steps have accumulated so far
            self._enqueue_step(batchX, batchY)
            l, _n = self._read_acc()
        summary = {"learning_rate": self.learningRateVal, "l2Lambda": self.l2RegularizationLambdaVal,
                   "loss1": l[0], "loss2": l[1], "loss3": l[2], "loss4": l[3], "lossL2": l[4], "loss": l[5]}
        return np.float32(l[5]), summary

    def train(self, batchX, batchY):
        """v3.py:183-193 -> (loss, summary)"""
        return self._train_step_impl(batchX, batchY)

    def trainNoRT(self, batchX, batchY):
        """v3.py:195-205"""
        self.trainLossRTVal = None; self.trainSummaryRTVal = None
        self.trainLossRTVal, self.trainSummaryRTVal = self._train_step_impl(batchX, batchY)

    def getLoss(self, batchX, batchY):
        """v3.py:207-216: phase False, dropout 0, lambda 0"""
        return self._loss_only(batchX, batchY)

    def getLossNoRT(self, batchX, batchY):
        """v3.py:218-227"""
        self.getLossLossRTVal = None
        self.getLossLossRTVal = self._loss_only(batchX, batchY)

    def setLearningRate(self, learningRate=None):
        """v3.py:229-234"""
        if learningRate is None:
            self.learningRateVal = self.learningRateVal * self.learningRateDecay
        else:
            self.learningRateVal = learningRate
        return self.learningRateVal

    def setL2RegularizationLambda(self, l2RegularizationLambda=None):
        """v3.py:236-241"""
        if l2RegularizationLambda is None:
            self.l2RegularizationLambdaVal = self.l2RegularizationLambdaVal * self.l2RegularizationLambdaDecay
        else:
            self.l2RegularizationLambdaVal = l2RegularizationLambda
        return self.l2RegularizationLambdaVal

    def saveParameters(self, fn):
        """v3.py:243-246 (tf.train.Saver.save): writes fn.index / fn.data-00000-of-00001 /
        fn.meta in the TensorFlow V2 checkpoint layout."""
        from . import tf_checkpoint
        tf_checkpoint.save_model(self, fn)

    def restoreParameters(self, fn):
        """v3.py:248-251 (tf.train.Saver.restore)"""
        from . import tf_checkpoint
        tf_checkpoint.restore_model(self, fn)

    def summaryFileWriter(self, logsPath):
        """v3.py:253-255: returns an object with add_summary(summary, step)."""
        from . import summary
        return summary.ScalarLogWriter(logsPath)
