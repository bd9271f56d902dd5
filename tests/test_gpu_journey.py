"""End-to-end user journey on the GPU, the way AttemptFour/main.py + eval.py drive the reference: config.yaml ->
build_model -> fit (callbacks, validation, checkpoint) -> load_weights in a fresh model -> eval dumps."""
import os

import numpy as np
import pytest
import yaml

pytestmark = pytest.mark.gpu


def test_config_fit_checkpoint_eval(tmp_path):
    from masters_thesis_amd import config as CFG, callbacks as CB, data as D, evaluate as EV
    from masters_thesis_amd.lc_nic import synthetic_groups
    cfg = dict(run="journey", log="./Log/", seed=42, epochs=2, batch_size=16, max_length=10, top_k=300, optimizer="Adam",
               alpha=0.0001, clipnorm=0.1, decay=0, dropout_input=0, dropout_features=0.2, dropout_text=0.2,
               dropout_lstm=0.2, dropout_attn=0.2, dropout_out=0.2, input_reg=0.01, attn_reg=0.001, lstm_reg=0.00003,
               output_reg=0.00001, units=64, attn_units=32, group_size=32, embedding_features=512, embedding_text=64)
    p = tmp_path / "config.yaml"
    p.write_text(yaml.dump(cfg))
    config = CFG.load_config(str(p))
    N, R, V = 3000, 40, config["top_k"] + 1
    groups = synthetic_groups(N, R, config["group_size"], seed=1)
    model = CFG.build_model(config, groups)
    B, T, U = config["batch_size"], config["max_length"], config["units"]
    train = D.SyntheticGenerator(8, B, N, U, T, V, seed=1, one_hot=True)       # one-hot targets, as the reference feeds them
    val = D.SyntheticGenerator(2, B, N, U, T, V, seed=9)
    cbs = [CB.LossHistory(str(tmp_path / "loss_history.csv")),
           CB.ModelCheckpoint(str(tmp_path / "model" / "model-ep{epoch:03d}.npz"), monitor="val_loss", save_best_only=False)]
    hist = model.fit(train, epochs=3, steps_per_epoch=8, batch_size=B, callbacks=cbs, validation_data=val,
                     validation_steps=2, initial_epoch=0, verbose=0)
    assert len(hist["loss"]) == 3 and np.isfinite(hist["loss"]).all() and np.isfinite(hist["val_loss"]).all()
    assert hist["loss"][-1] < hist["loss"][0]                                  # 24 Adam steps on 8 recurring batches
    assert {"loss", "L2", "accuracy", "attention", "val_loss", "val_accuracy"} <= set(hist)
    ck = tmp_path / "model" / "model-ep003.npz"
    assert ck.exists()
    m2 = CFG.build_model(config, groups)
    m2.load_weights(str(ck), by_name=True, skip_mismatch=True)
    batch = val[0]
    a, b = model.test_step(batch).as_floats(), m2.test_step(batch).as_floats()
    assert a == b
    # eval.py: greedy decode of the generator, dumps in the reference's layouts
    tok = D.Tokenizer(num_words=V, oov_token="<unk>")
    tok.fit_on_texts(["<start> a b c <end>"])
    tok.word_index["<start>"] = 1
    outs, attn = EV.eval_model(m2, val, tok, config, str(tmp_path / "eval"), 3)
    assert outs.shape == (2 * B, T, 1) and attn.shape == (2 * B, T, R, 1)
    assert os.path.exists(tmp_path / "eval" / "output_captions_raw_3.npy")
    assert np.abs(attn.sum(axis=2) - 1).max() < 1e-5                           # attention maps are distributions over regions
