"""silent_speech_amd -- MI355X (gfx950) implementation of the EMG->mel transduction training hot
path of dgaddy/silent_speech, behind the reference's own Python API."""
__version__ = '0.1.0'
