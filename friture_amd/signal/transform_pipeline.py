"""friture/signal/transform_pipeline.py:23-34: a list of blocks, each with push(data) -> data."""


class Transform_Pipeline:
    def __init__(self, blocks):
        self.blocks = blocks

    def push(self, data):
        for block in self.blocks:
            data = block.push(data)
        return data
