"""Feature front-end: waveform -> log-mel spectrogram + energy (mirror of the reference's `audio` package surface that the
preprocessor uses: `Audio.stft.TacotronSTFT`, `Audio.tools.get_mel_from_wav`)."""
from . import stft, tools  # noqa: F401
