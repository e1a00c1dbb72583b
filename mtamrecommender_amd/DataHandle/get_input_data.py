"""Batch iterator.  The reference's ``DataHandle.get_input_data.DataInput`` is not
in its tree; its use at train_process.py:240,326 --
``for step_i, batch in DataInput(data, batch_size)`` -- fixes the contract:
sequential, non-overlapping slices, a short final batch, 1-based step index."""


class DataInput(object):

    def __init__(self, data, batch_size):
        self.data = data
        self.batch_size = int(batch_size)
        self.epoch_size = len(data) // self.batch_size
        if self.epoch_size * self.batch_size < len(data):
            self.epoch_size += 1
        self.i = 0

    def __iter__(self):
        return self

    def __next__(self):
        if self.i == self.epoch_size:
            raise StopIteration
        batch = self.data[self.i * self.batch_size:min((self.i + 1) * self.batch_size, len(self.data))]
        self.i += 1
        return self.i, batch

    next = __next__
