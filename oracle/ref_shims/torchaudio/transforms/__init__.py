class Spectrogram:  # names only; the data pipeline is never executed
    pass


class MelScale:
    pass


class MelSpectrogram:
    pass


class Resample:
    pass
