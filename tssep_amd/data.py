"""DummyReader -- drop-in for tssep/data.py:11-152 (deterministic synthetic 8-speaker data).

Behaviour (regenerated bit-exactly, tests/golden/dummy_reader.npz):
  * an example is fully determined by its seed: ``RandomState(seed)`` first draws 3 integer tone
    frequencies in [100, 7000) per speaker (one ``randint`` of shape [3, speakers]), then one uniform noise
    row of N samples (data.py:75-104);
  * a speaker's clean signal is the float32 sum of its three unit sinusoids, gated by the staircase
    activity of ``_get_vad`` (neighbours overlap by half, data.py:34-56); the observation is the sum of
    the gated signals plus the noise row;
  * the speaker "embedding" marks, for every tone, the two neighbouring cells
    ``[f * aux_size // 7001, +2)`` of an ``aux_size``-long zero vector (data.py:108-118);
  * 'train' datasets hold ``train_examples`` examples, every other dataset 4, seeds counting from 0.
"""
import dataclasses

import numpy as np

from . import dataset as lazy_dataset
from .configurable import Configurable

_TONE_RANGE = (100, 7000)        # Hz, upper bound exclusive
_TONES_PER_SPEAKER = 3


def _staircase(num_samples, num_speakers):
    """Speaker i is active on [start_i, end_i): end_i = N (i + 2) // (S + 1), and the next speaker
    starts in the middle of the current one's stretch."""
    active = np.zeros((num_speakers, num_samples), dtype=bool)
    begin = 0
    for spk in range(num_speakers):
        stop = num_samples * (spk + 2) // (num_speakers + 1)
        active[spk, begin:stop] = True
        begin = stop - (stop - begin) // 2
    return active


def _tone_cells(tones, aux_size):
    """[tones, speakers] integer frequencies -> [speakers, aux_size] float32 with 1 in the two cells
    starting at ``f * aux_size // (f_max + 1)`` of every tone (clipped at the end of the vector)."""
    cells = (tones.T * aux_size) // (_TONE_RANGE[1] + 1)               # [speakers, tones]
    aux = np.zeros((tones.shape[1], aux_size), dtype=np.float32)
    spk = np.repeat(np.arange(tones.shape[1]), tones.shape[0])
    for shift in (0, 1):
        col = cells.reshape(-1) + shift
        keep = col < aux_size
        aux[spk[keep], col[keep]] = 1
    return aux


@dataclasses.dataclass
class DummyReader(Configurable):
    train_dataset_name: str = "train"
    validate_dataset_name: str = "validate"
    domain_adaptation_src_dataset_name: str = "validate"
    eval_dataset_name: str = "eval"
    sample_rate: int = 16000
    aux_size: int = 100
    train_examples: int = 10

    def _get_vad(self, num_samples, num_speakers):
        return _staircase(num_samples, num_speakers)

    def get_example(self, seed, dataset_name, load_keys=("observation", "speaker_reverberation_early_ch0",
                                                         "vad"), num_speakers=8, seconds=5):
        n = self.sample_rate * seconds
        rng = np.random.RandomState(seed)
        tones = rng.randint(*_TONE_RANGE, size=(_TONES_PER_SPEAKER, num_speakers))
        noise = None            # drawn AFTER the tones, below: the order of the draws is the contract
        t = np.arange(n) / self.sample_rate
        # float64 phase 2*pi*f*t evaluated as ((2 pi) f) t, summed over the tones, then rounded once
        clean = np.sin(2 * np.pi * tones[..., None] * t).sum(axis=0).astype(np.float32)
        vad = self._get_vad(n, num_speakers)
        clean = clean * vad
        noise = rng.rand(1, n).astype(np.float32)
        audio = {"observation": clean[:, None, :].sum(axis=0) + noise, "vad": vad}
        if "speaker_reverberation_early_ch0" in load_keys:
            audio["speaker_reverberation_early_ch0"] = clean
        return {"example_id": f"dummy_id_{seed}", "num_samples": n, "audio_data": audio,
                "auxInput": _tone_cells(tones, self.aux_size), "dataset": dataset_name}

    def __call__(self, dataset_name, pre_load_apply=None, load_keys=("observation",)):
        count = self.train_examples if "train" in dataset_name else 4
        ds = lazy_dataset.new([self.get_example(seed, dataset_name, load_keys) for seed in range(count)])
        return ds if pre_load_apply is None else pre_load_apply(ds)

    class data_hooks:
        """tssep/data.py:148-152: the hook point before the network; the toy reader needs nothing."""

        @staticmethod
        def pre_net(ex):
            return ex
