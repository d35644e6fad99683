"""fairseq/data/dictionary.py subset: symbol table with bos=0 pad=1 eos=2 unk=3, `load` from a fairseq dict file
(e.g. chimera/resources/wmt14-en-de-spm/spm_unigram10000_wave_joint.txt: 9 996 lines + 4 specials = 10 000)."""


class Dictionary:
    def __init__(self, bos="<s>", pad="<pad>", eos="</s>", unk="<unk>"):
        self.symbols, self.count, self.indices = [], [], {}
        self.bos_word, self.pad_word, self.eos_word, self.unk_word = bos, pad, eos, unk
        self.bos_index = self.add_symbol(bos)
        self.pad_index = self.add_symbol(pad)
        self.eos_index = self.add_symbol(eos)
        self.unk_index = self.add_symbol(unk)
        self.nspecial = len(self.symbols)

    def __len__(self):
        return len(self.symbols)

    def __getitem__(self, idx):
        return self.symbols[idx] if idx < len(self.symbols) else self.unk_word

    def index(self, sym):
        return self.indices.get(sym, self.unk_index)

    def add_symbol(self, word, n=1, overwrite=False):
        if word in self.indices and not overwrite:
            idx = self.indices[word]
            self.count[idx] += n
            return idx
        idx = len(self.symbols)
        self.indices[word] = idx
        self.symbols.append(word)
        self.count.append(n)
        return idx

    def bos(self):
        return self.bos_index

    def pad(self):
        return self.pad_index

    def eos(self):
        return self.eos_index

    def unk(self):
        return self.unk_index

    def string(self, tensor, bpe_symbol=None, escape_unk=False):
        toks = [self[int(i)] for i in tensor if int(i) not in (self.eos_index, self.pad_index)]
        return " ".join(toks)

    def encode_line(self, line, add_if_not_exist=True, append_eos=True, reverse_order=False):
        """data/dictionary.py:292-317 with tokenizer.tokenize_line (:12-15): whitespace split -> IntTensor of indices (+ eos)."""
        import re

        import torch
        words = re.sub(r"\s+", " ", line).strip().split()
        if reverse_order:
            words = list(reversed(words))
        ids = torch.IntTensor(len(words) + 1 if append_eos else len(words))
        for i, w in enumerate(words):
            ids[i] = self.add_symbol(w) if add_if_not_exist else self.index(w)
        if append_eos:
            ids[len(words)] = self.eos_index
        return ids

    @classmethod
    def load(cls, f):
        d = cls()
        with open(f, "r", encoding="utf-8") as fd:
            for line in fd:
                line = line.rstrip()
                if not line:
                    continue
                word, cnt = line.rsplit(" ", 1)
                d.add_symbol(word, n=int(cnt), overwrite="#fairseq:overwrite" in line)
        return d

    @classmethod
    def synthetic(cls, n):
        """n symbols incl. the 4 specials (10 000 for the shipped SPM dictionary)."""
        d = cls()
        for i in range(n - d.nspecial):
            d.add_symbol("w%d" % i)
        return d
