"""Stand-in for tf.summary.FileWriter (clairvoyante_v3.py:253-255, train.py:55,115-116):
the reference logs the scalars learning_rate, l2Lambda, loss1-4, lossL2, loss per batch,
stamped with the epoch index.  Written as tab-separated text (TensorBoard event files
are out of scope)."""
import os


class ScalarLogWriter(object):
    KEYS = ("learning_rate", "l2Lambda", "loss1", "loss2", "loss3", "loss4", "lossL2", "loss")

    def __init__(self, logsPath):
        os.makedirs(logsPath, exist_ok=True)
        self._fh = open(os.path.join(logsPath, "scalars.tsv"), "a")
        if self._fh.tell() == 0:
            self._fh.write("step\t" + "\t".join(self.KEYS) + "\n")

    def add_summary(self, summary, step):
        if summary is None:
            return
        self._fh.write(str(step) + "\t" + "\t".join("%.9g" % summary.get(k, float("nan")) for k in self.KEYS) + "\n")

    def flush(self):
        self._fh.flush()

    def close(self):
        self._fh.close()
