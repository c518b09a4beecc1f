"""Stream constants of the audio front-end (friture/audiobackend.py:31-32)."""
SAMPLING_RATE = 48000
FRAMES_PER_BUFFER = 512
NOCTAVE = 9          # friture/filter.py:7
FIR_LENGTH = 512     # friture/octavefilters.py:35
