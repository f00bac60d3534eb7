"""librosa is absent here.  Only librosa.filters.mel is needed by the reference's mel_spectrogram
(data_utils.py:47); it is served from the oracle's Slaney restatement (oracle/mel_ref.py), so the
mel BASIS is 'parity unpinned' while everything downstream of it is pinned.  Fixture tooling only."""
from . import filters
