"""A chain of processing stages with the block protocol of friture/signal/transform_pipeline.py:23-34:
every stage exposes push(columns) -> columns, and the chain itself is such a stage.  The spectrogram
widget builds it from the frequency resampler, the online time resampler and the colour transform
(friture/spectrogram.py:62-68) and reaches into `.blocks` to reconfigure them."""
from functools import reduce


class Transform_Pipeline:
    def __init__(self, blocks):
        self.blocks = blocks

    def push(self, data):
        return reduce(lambda columns, stage: stage.push(columns), self.blocks, data)
