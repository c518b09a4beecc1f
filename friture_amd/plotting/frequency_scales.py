"""Frequency scales of the spectrogram's vertical axis: `transform` / `inverse` pairs with the
class names of friture/plotting/frequency_scales.py:65-293 (tick generation is drawing code and
stays with the reference's plotting layer).  These map the screen rows of Frequency_Resampler to
frequencies once per geometry change; they are plan constants, evaluated on the host."""
import numpy as np


class Linear:
    NAME = 'Linear'

    @staticmethod
    def transform(frequency):
        return frequency

    @staticmethod
    def inverse(value):
        return value


class Logarithmic:
    NAME = 'Logarithmic'

    @staticmethod
    def transform(frequency):
        return np.log10(frequency)

    @staticmethod
    def inverse(logs):
        return 10 ** logs


class Mel:
    NAME = 'Mel'

    @staticmethod
    def transform(frequency):
        return 2595 * np.log10(1 + frequency / 700)

    @staticmethod
    def inverse(mels):
        return 700 * (10 ** (mels / 2595) - 1)


class Erb:
    NAME = 'ERB'
    A = 21.33228113095401739888262

    @staticmethod
    def transform(frequency):
        return Erb.A * np.log10(1 + 0.00437 * frequency)

    @staticmethod
    def inverse(erbs):
        return (10 ** (erbs / Erb.A) - 1) / 0.00437


class Octave:
    NAME = 'Octave'

    @staticmethod
    def transform(frequency):
        return np.log2(np.fmax(frequency, 1e-20))

    @staticmethod
    def inverse(logs):
        return 2 ** logs


class OctaveC(Octave):
    """The log2 axis whose major ticks sit on the C of every octave (friture/plotting/frequency_scales.py:225-237): same
    transform pair as `Octave`, its own scale name."""
    NAME = 'OctaveC'


ALL = [Linear, Logarithmic, Mel, Erb, Octave, OctaveC]
