"""The part of friture/plotting the spectrogram pipeline needs: frequency scale transforms."""
