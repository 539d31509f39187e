"""Synthetic data -- oracle restatement of tssep/data.py:34-146 (DummyReader)."""
import numpy as np


def get_vad(num_samples, num_speakers):
    """tssep/data.py:49-56 staircase VAD with 50 % neighbour overlap."""
    vad = np.zeros((num_speakers, num_samples), dtype=bool)
    start = 0
    for i in range(num_speakers):
        end = num_samples * (i + 2) // (num_speakers + 1)
        vad[i, start:end] = True
        start = end - (end - start) // 2
    return vad


def dummy_example(seed, sample_rate=16000, aux_size=100, num_speakers=8,
                  seconds=5, dataset="validate"):
    """tssep/data.py:74-139 (sinous=True branch), one channel."""
    num_samples = sample_rate * seconds
    rng = np.random.RandomState(seed)
    max_frequency, min_frequency, num_frequencies = 7000, 100, 3
    frequency = rng.randint(min_frequency, max_frequency,
                            size=(num_frequencies, num_speakers))
    time = np.arange(num_samples) / sample_rate
    early = np.sin(2 * np.pi * frequency[..., None] * time).sum(axis=0) \
        .astype(np.float32)[:, None, :]
    vad = get_vad(num_samples, num_speakers)
    early = early * vad[:, None, :]
    noise = 1 * rng.rand(1, num_samples).astype(np.float32)
    observation = early.sum(axis=0) + noise
    aux = np.zeros((num_speakers, aux_size), dtype=np.float32)
    scale = max_frequency + 1
    for spk, fs in enumerate(frequency.T):
        for f in fs:
            f = (f * aux.shape[1]) // scale
            aux[spk, f:f + 2] = 1
    return dict(example_id=f"dummy_id_{seed}", num_samples=num_samples,
                observation=observation,
                speaker_reverberation_early_ch0=early[:, 0],
                vad=vad, auxInput=aux, dataset=dataset)
