"""Constants of the experiment layout — values identical to the reference's properties.py
(dataset directory names :1-20, checkpoint/out dirs :23-29, input_size :33, 95-symbol char_set
:35-36 with blank '`' at index 0, empty_char :40, max_char_len :41)."""
import string

_DATASETS = {
    "pos_text": "textarea_dataset",   # POS text areas
    "vgg_text": "vgg",                # VGG synthetic words
    "patch": "patch_dataset",         # POS document patches
    "wr": "wildreceipt",              # WildReceipt patches
}
pos_text_dataset_train, pos_text_dataset_test, pos_text_dataset_dev = (f"{_DATASETS['pos_text']}_{s}" for s in ("train", "test", "dev"))
vgg_text_dataset_train, vgg_text_dataset_test, vgg_text_dataset_dev = (f"{_DATASETS['vgg_text']}_{s}" for s in ("train", "test", "dev"))
patch_dataset_train, patch_dataset_test, patch_dataset_dev = (f"{_DATASETS['patch']}_{s}" for s in ("train", "test", "dev"))
wr_dataset_train, wr_dataset_test, wr_dataset_dev = (f"{_DATASETS['wr']}_{s}" for s in ("train", "test", "dev"))

cer_artifacts_path = "cer_artifacts"
prep_crnn_ckpts = "ckpts"
crnn_model_path = "./outputs/crnn_trained_model/model"
crnn_tensor_board = "./outputs/crnn_runs/"
prep_model_path = "./outputs/prep_trained_model/"
img_out = "img_out"
param_path = "params.txt"
train_subset_size = 50000
val_subset_size = 10000

input_size = (32, 128)
num_workers = 4

# index 0 is the CTC blank; order is part of every CRNN checkpoint's meaning
char_set = (["`", " "] + list("!\"#$%&'()*+,-.") + list(string.digits) + list(":;<=>?@") + list(string.ascii_uppercase)
            + list("[]^") + list(string.ascii_lowercase) + list("{|~") + ["€", "}", "\\", "/"])
assert len(char_set) == 95

tesseract_path = ""
empty_char = " "
max_char_len = 100
