"""``config.yaml`` loader: the reference's YAML (AttemptFour/config.yaml:1-60, loaded with
yaml.safe_load at main.py:36-38) is accepted unchanged; ``build_model`` constructs the model
the way main.py:113-134 does."""
import yaml

from .optimizers import Adam, SGD, CategoricalCrossentropy


def load_config(path):
    with open(path, "r") as f:
        return yaml.safe_load(f)


def build_optimizer(config):
    """main.py:96-104 (the Adam learning rate is hard-coded to 1e-4 there; config['alpha'] holds the same value)."""
    if config["optimizer"] == "Adam":
        return Adam(learning_rate=0.0001, beta_1=0.9, beta_2=0.98, epsilon=10.0e-9, clipnorm=config["clipnorm"])
    if config["optimizer"] == "SGD":
        return SGD(learning_rate=config["alpha"], momentum=0.9, nesterov=False)
    raise ValueError("No optimizer specified")


def build_model(config, groups, mode="attention", input_size=None, **kw):
    """lc_NIC.NIC(...) exactly as main.py:113-132 builds it, compiled as at main.py:134.
    mode="fc" builds the fully-connected variant instead (the call_fc / greedy_predict_fc dispatch the
    reference selects by editing lc_NIC.py:163-167,507-509); it needs ``input_size`` (#voxels)."""
    vocab_size = config["top_k"] + 1
    if mode == "fc":
        from .fc_nic import NICfc
        model = NICfc(input_size, config["units"], config["embedding_features"], config["embedding_text"], vocab_size,
                      config["max_length"], config["dropout_input"], config["dropout_features"], config["dropout_text"],
                      config["dropout_lstm"], config["dropout_out"], config["input_reg"], config["lstm_reg"],
                      config["output_reg"], seed=config.get("seed", 42), **kw)
    elif mode == "attention":
        from .lc_nic import NIC
        model = NIC(groups, config["units"], config["embedding_features"], config["embedding_text"],
                    config["attn_units"], vocab_size, config["max_length"], config["dropout_input"],
                    config["dropout_features"], config["dropout_text"], config["dropout_attn"], config["dropout_lstm"],
                    config["dropout_out"], config["input_reg"], config["attn_reg"], config["lstm_reg"],
                    config["output_reg"], seed=config.get("seed", 42), **kw)
    else:
        raise ValueError(f"unknown mode {mode!r}")
    model.compile(build_optimizer(config), CategoricalCrossentropy(from_logits=False, reduction="none"), run_eagerly=True)
    return model
